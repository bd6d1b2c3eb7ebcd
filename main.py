#!/usr/bin/env python3
"""`python main.py --model=... --inference=... --method=...` -- the reference's CLI
(main.py) on the MI355X engine; see autoreparam_amd/main.py."""
import sys

from autoreparam_amd.main import main

if __name__ == "__main__":
    main(sys.argv[1:])
