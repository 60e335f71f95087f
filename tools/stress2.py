"""Derived-logD context vs a stored-logD context holding the same logD: must stay bit-identical.  Checks every sweep:
labels, size tables, and the row sums of a few clusters (which generation of the S table goes wrong first?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K, sweeps = 2048, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 1500
kern = sys.argv[2] if len(sys.argv) > 2 else "auto"
data = rc.generatemixture(n, K, seed=5, sigma=0.6, dim=6)
sh = np.random.default_rng(8).permutation(n)
D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)]); truth = data["clusts"][sh]
P = dict(rc.likelihood_hyperparams(D, truth), maxK=12)
A = rc.Context(D, kcap=64); A.set_params(**P); A.set_bulk_kernel(kern); A.set_state(truth)
L = A.get_matrix(1)
B = rc.Context(D, logD=L, kcap=64); B.set_params(**P); B.set_bulk_kernel(kern); B.set_state(truth)
sh_L = B.debug_rowsums(int(truth[0]))[3] - A.debug_rowsums(int(truth[0]))[3]   # stored mode may use a finer grid (exact multiples)
assert sh_L >= 0
for t in range(sweeps):
    A.gibbs_sweep(1.0, 0.5, 3, t); B.gibbs_sweep(1.0, 0.5, 3, t)
    a, b = A.get_state(), B.get_state()
    ca = np.bincount(a[0], minlength=n + 1)[1:]
    rs_bad = []
    for lab in np.unique(b[0])[:6]:
        ra, rb = A.debug_rowsums(int(lab)), B.debug_rowsums(int(lab))
        if not (np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1] << sh_L, rb[1])):
            rs_bad.append((int(lab), int(np.sum(ra[0] != rb[0])), int(np.sum((ra[1] << sh_L) != rb[1]))))
    eq, okA = np.array_equal(a[0], b[0]), np.array_equal(a[1], ca)
    if rs_bad or not eq or not okA:
        print(f"t={t}: labels_equal={eq} A_consistent={okA} rowsum mismatches (label, #D, #L): {rs_bad} statsA={A.sweep_stats()} statsB={B.sweep_stats()}")
        break
print("done", t)
