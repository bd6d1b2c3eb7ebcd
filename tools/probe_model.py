"""One model / lanes-per-chain through the fused HMC kernel (for rocprofv3 --pmc runs):
   probe_model.py <election|radon_MN|radon_PA|8schools|electric|german> C L lanes reparam [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import model_sweep as ms
import helpers
name, C, L, lanes, rep = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
T = int(sys.argv[6]) if len(sys.argv) > 6 else 16
eps = {"election": 0.005, "german": 0.005, "electric": 0.01}.get(name, 0.05)
ms.run(name, helpers.spec(name), C, L, lanes, rep, T=T, eps=eps)
