"""iterations/s of rc_run_chain with the reference's default options on the reference's own example data (paper dataset 1, n = 100,
random initial labels, accepted proposals kept): pipelined against synchronous loop (bench.py's with_accepted_proposals leg alone)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = np.load(os.path.join(ROOT, "tests", "golden", "paper_datasets.npz"))
D1, lab1 = np.ascontiguousarray(z["D1"]), z["labels1"]
P1 = rc.likelihood_hyperparams(D1, lab1)
init1 = np.random.default_rng(1).integers(1, 11, size=100).astype(np.int64)
for thin in (10, 1):
    for name, env in (("pipelined", None), ("synchronous", "0")):
        if env is None: os.environ.pop("RC_CHAIN_PIPELINE", None)
        else: os.environ["RC_CHAIN_PIPELINE"] = env
        cr = rc.Context(D1); cr.set_params(**P1); cr.set_state(init1); cr.cocluster_reset(); cr.attach_host_matrices(D1)
        cr.run_chain(200, 0, thin, 5, 1, 3, 1.0, 0.5, 1.0, splitmerge="intended")
        its = 3000
        t1 = time.perf_counter()
        ch = cr.run_chain(its, 0, thin, 5, 1, 3, 1.0, 0.5, 1.0, splitmerge="intended", first_iter=200)
        dt = time.perf_counter() - t1
        cs = cr.chain_stats()
        print(f"thin={thin} {name}: {its / dt:8.1f} it/s  acceptances {int(ch['splitmerge_acceptances'].sum())} splits {int(ch['splitmerge_splits'].sum())} rollbacks {cs['rollbacks']} workers {cs['workers']} K {int(ch['K'][-1])}")
        cr.close()
os.environ.pop("RC_CHAIN_PIPELINE", None)
