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
