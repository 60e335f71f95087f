"""A short batch of the randomised differential check (tests/fuzz_parity.py: random sizes, overlaps, capacities, batch capacities,
storage widths, stored / derived logD, maxK, repulsion, stream arrangement, modes; five sweeps from random labels against the
oracle; speculative against synchronous chain loop).  By hand over round 2: 7100 small and 450 large (4100 ≤ n < 7000: symmetric kernels,
re-layouts) sweep cases — the later 5400 with the resolver's score cache off / always on / adaptive at random, 400 of them on the chaos build (block-dependent random delays inside the resolver's rounds) —, 1370 speculative-vs-synchronous chain cases and 2450 chain-vs-oracle-loop cases without a mismatch — after the chain comparison had found last-bit differences of
loglik in 6 of 1000 chains (slot-order summation, fixed)."""
import pytest

pytestmark = pytest.mark.gpu


def test_randomised_sweeps_against_the_oracle():
    import fuzz_parity
    assert fuzz_parity.run(40, 5000) == 0


def test_randomised_chains_speculative_against_synchronous():
    import fuzz_parity
    assert fuzz_parity.run_chains(8, 9000) == 0


def test_randomised_chains_against_the_oracle_loop():
    import fuzz_parity
    assert fuzz_parity.run_chains_oracle(12, 11000) == 0


def test_randomised_wide_contexts_against_the_oracle():
    """More than 4096 clusters (round 4: k_sweep_wide; capacities that leave a wide context room to overflow again — the case that
    dropped sweeps silently until the large leg of these checks met it): three cases, three sweeps each, labels / sizes / K / change
    counts exact, log-likelihood 1e-9, recorded sample exact.  By hand: 160 cases, 30 of them on the chaos build."""
    import fuzz_parity
    assert fuzz_parity.run_wide(3, 84000) == 0
