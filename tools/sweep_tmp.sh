python -m pytest tests -m gpu -x -q -k "ess or cli" 2>&1 | tail -3
python tools/ess_breakdown.py 8 | tail -1
python tools/ess_breakdown.py 2 | tail -1
