import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import ess_ref
from autoreparam_amd import util
gpu = torch.device("cuda:0")
def both(x, **kw):
    os.environ["ARP_DEBUG"] = "1"
    os.environ["ARP_ESS_ONEPASS"] = "1"; a = util.effective_sample_size(x, **kw).cpu().numpy()
    os.environ["ARP_ESS_ONEPASS"] = "0"; b = util.effective_sample_size(x, **kw).cpu().numpy()
    return a, b
# ragged width: 2 waves + 44 series, mixed rho
S, Cn, D = 700, 60, 5
rho = np.array([0.0, 0.3, 0.6, 0.9, -0.4])
x64 = ess_ref.ar1(S, (Cn, D), rho, seed=5) * [1.0, 10.0, 0.1, 3.0, 1.0] + [0.0, 100.0, -5.0, 1e3, 0.0]
x = torch.as_tensor(x64, dtype=torch.float32); xd = x.to(gpu)
a, b = both(xd); ref = ess_ref.ess_fft(x.numpy())
print("ragged 300: max rel one-pass vs oracle %.2e, two-sweep vs oracle %.2e" % (np.abs(a/ref-1).max(), np.abs(b/ref-1).max()))
# odd width, strided view
a2, b2 = both(xd[:, 3:40, :]); print("strided 185: equal to full:", np.array_equal(a2, a[3:40]), np.abs(a2/ref[3:40]-1).max())
# short series
for s_ in (9, 30, 49, 100):
    sh = x[:s_].contiguous(); a3, b3 = both(sh.to(gpu)); r3 = ess_ref.ess_fft(sh.numpy())
    print("S=%d: one-pass %.2e two-sweep %.2e" % (s_, np.nanmax(np.abs(a3/r3-1)), np.nanmax(np.abs(b3/r3-1))))
# drift
drift = x.clone(); drift[:20] += torch.tensor([50.0, 5e3, 3.0, 2e4, -80.0])
a4, b4 = both(drift.to(gpu)); r4 = ess_ref.ess_fft(drift.numpy()); print("drift: %.2e %.2e" % (np.abs(a4/r4-1).max(), np.abs(b4/r4-1).max()))
# constant
c = torch.ones(50, 40, 5, device=gpu); a5, _ = both(c); print("constant all nan:", np.isnan(a5).all())
# slow series, cooperative tail (S + 72 <= 2304)
slow = torch.as_tensor(ess_ref.ar1(2000, (50, 3), [0.98, 0.95, 0.5], seed=7), dtype=torch.float32)
a6, b6 = both(slow.to(gpu)); r6 = ess_ref.ess_fft(slow.numpy()); print("slow S=2000: %.2e %.2e" % (np.abs(a6/r6-1).max(), np.abs(b6/r6-1).max()))
# long series: workspace (matrix-core tail) and none (far sweeps)
lg = torch.as_tensor(ess_ref.ar1(6000, (30, 5), [0.995, 0.97, 0.5, 0.9, -0.2], seed=11) * [1.0, 4.0, 0.1, 30.0, 1.0] + [0.0, -20.0, 5.0, 1e3, 0.0], dtype=torch.float32)
a7, b7 = both(lg.to(gpu)); r7 = ess_ref.ess_fft(lg.numpy()); print("long S=6000 ws: %.2e %.2e" % (np.abs(a7/r7-1).max(), np.abs(b7/r7-1).max()))
import ctypes as C
from autoreparam_amd import _lib
L = _lib.lib(); out = torch.empty(30, 5, device=gpu); ld = lg.to(gpu)
os.environ["ARP_ESS_ONEPASS"] = "1"
_lib.check(L.arp_ess(C.c_void_p(ld.data_ptr()), 6000, 150, 150, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
print("long S=6000 no ws: %.2e" % np.abs(out.cpu().numpy()/r7-1).max())
# trend
t = torch.linspace(0, 1, 2500, device=gpu).reshape(-1, 1, 1) + 0.01 * torch.randn(2500, 50, 3, device=gpu)
a8, b8 = both(t); r8 = ess_ref.ess_fft(t.cpu().numpy()); print("trend: %.2e %.2e" % (np.abs(a8/r8-1).max(), np.abs(b8/r8-1).max()))
