"""Two contexts on the same moving chain (default kernel selection vs forced perm kernel); every few sweeps both
states are pulled and checked: equal labels, and each context's size table consistent with its own labels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K, sweeps, every = 2048, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 1500, 3
data = rc.generatemixture(n, K, seed=5, sigma=0.6, dim=6)
sh = np.random.default_rng(8).permutation(n)
D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)]); truth = data["clusts"][sh]
P = dict(rc.likelihood_hyperparams(D, truth), maxK=12)
A = rc.Context(D, kcap=64); A.set_params(**P); A.set_state(truth)
B = rc.Context(D, kcap=64); B.set_params(**P); B.set_bulk_kernel("perm"); B.set_state(truth)
bad = 0
for t in range(sweeps):
    A.gibbs_sweep(1.0, 0.5, 3, t, blocking=False); B.gibbs_sweep(1.0, 0.5, 3, t, blocking=False)
    if t % every == every - 1:
        a, b = A.get_state(), B.get_state()
        ca, cb = np.bincount(a[0], minlength=n + 1)[1:], np.bincount(b[0], minlength=n + 1)[1:]
        okA, okB, eq = np.array_equal(a[1], ca), np.array_equal(b[1], cb), np.array_equal(a[0], b[0])
        if not (okA and okB and eq):
            bad += 1
            print(f"t={t}: labels_equal={eq} A_consistent={okA} B_consistent={okB} statsA={A.sweep_stats()} statsB={B.sweep_stats()}")
            if not okA: print("   A: idx", np.flatnonzero(a[1] != ca), "summary", a[1][a[1] != ca], "counted", ca[a[1] != ca])
            if not okB: print("   B: idx", np.flatnonzero(b[1] != cb), "summary", b[1][b[1] != cb], "counted", cb[b[1] != cb])
            if bad >= 4: break
print("done: inconsistencies", bad)
