"""A short batch of the randomised differential check (tests/fuzz_parity.py: random sizes, overlaps, capacities, batch capacities,
storage widths, stored / derived logD, maxK, repulsion, stream arrangement, modes; five sweeps from random labels against the
oracle).  1350 further cases were run once by hand with no mismatch."""
import pytest

pytestmark = pytest.mark.gpu


def test_randomised_sweeps_against_the_oracle():
    import fuzz_parity
    assert fuzz_parity.run(40, 5000) == 0


def test_randomised_chains_speculative_against_synchronous():
    import fuzz_parity
    assert fuzz_parity.run_chains(8, 9000) == 0
