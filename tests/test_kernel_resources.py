"""Register, LDS and scratch budgets of the kernels whose co-residency the design rests on, read from the gfx950 ISA the
compiler emits (hipcc cross-compiles without a GPU): the resolver is a 128-register kernel that has to fit on a SIMD beside three waves
of the row reduction (4 × 128 = 512 registers per lane), the reductions must not spill (a scratch reload inside the streaming loop waits
for every prefetched row), and their LDS footprints are what `finish_create` sizes the resolver's tables against."""
import os, re, shutil, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "redclust.jl_amd", "csrc", "redclust_hip.hip")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "rc.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, SRC],
                   check=True, cwd=os.path.dirname(SRC), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    res, name, cur = {}, None, {}
    for line in open(out):
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            name, cur = m.group(1), {}
        for key in ("next_free_vgpr", "group_segment_fixed_size", "private_segment_fixed_size"):
            m = re.match(r"\s*\.amdhsa_" + key + r"\s+(\d+)", line)
            if m and name:
                cur[key] = int(m.group(1))
        if ".end_amdhsa_kernel" in line and name:
            res[name] = cur; name = None
    return res


def one(kernels, fragment):
    hits = [k for k in kernels if fragment in k]
    assert len(hits) == 1, (fragment, hits)
    return kernels[hits[0]]


def test_resolver_fits_beside_three_reduction_waves(kernels):
    k = one(kernels, "k_resolve4View")
    assert k["next_free_vgpr"] <= 128
    assert k["private_segment_fixed_size"] <= 256      # (a few spilled scalars of the prologue: 212 B in round 4; more means the loop spills)


@pytest.mark.parametrize("fragment,vgpr,lds", [
    ("k_bulk_syml2ILb1ELb1E", 128, 40960),     # the default path: three 40 KiB blocks per CU beside the resolver
    ("k_bulk_syml2ILb1ELb0E", 128, 40960),
    ("k_bulk_sym32ILi16E", 128, 36864),        # config 5: three 36 KiB blocks per CU beside the resolver
    ("k_bulkIxLb1E", 128, 65536),              # the full-read kernel of small problems and fragmented layouts
    ("k_bulkIxLb0E", 128, 65536),
])
def test_row_reductions_keep_their_budgets_and_do_not_spill(kernels, fragment, vgpr, lds):
    k = one(kernels, fragment)
    assert k["next_free_vgpr"] <= vgpr, k
    assert k["group_segment_fixed_size"] <= lds, k
    assert k["private_segment_fixed_size"] == 0, k


def test_block_tiled_reduction_leaves_room_for_the_resolver(kernels):
    k = one(kernels, "k_bulk_symILb0E")      # 64-bit storage, the caller's logD: two 68 KiB blocks per CU, 2 × 192 + 128 registers
    assert k["next_free_vgpr"] <= 192 and k["group_segment_fixed_size"] <= 69632 and k["private_segment_fixed_size"] == 0, k
