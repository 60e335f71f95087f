"""Cost of the 64-bit atomic flushes of k_bulk_syml<true>: the kernel alone (blocking sweeps, resolver commits disabled with
RC_DEBUG_FLAGS=2 so that the labels stay put), timed with HIP events, for the product build and for a build that computes
every flush but issues none (-DRC_EXP_NO_FLUSH).  usage: RC_DEBUG_FLAGS=2 [RC_LIB_PATH=build_exp/lib_noflush.so] python3 tools/flush_cost.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, redclust_amd as rc
n, K = 8192, 50
d = rc.generatemixture(n, K, seed=1); D, t = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, t)
c = rc.Context(D, kcap=128); c.set_params(**P); c.set_state(t); c.set_bulk_kernel("sym")
for s in range(10): c.gibbs_sweep(1.0, 0.5, 1, s)
c.kernel_timing(enable=1)
for s in range(10, 110): c.gibbs_sweep(1.0, 0.5, 1, s)
ms, launches = c.kernel_timing(enable=0)
print(f"{os.environ.get('RC_LIB_PATH', 'product build')}: {c.bulk_kernel_name()} alone {ms / launches * 1e3:.1f} us per launch ({launches} launches)")
