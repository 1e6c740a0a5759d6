"""The emitted ISA of the matrix-core backward pass holds what its hand-counted wait relies on (VERDICT r4 task 6).

csrc/bp_mfma.hpp prefetches the next knot's operands into LDS with `buffer_load_dword ... lds` and, at the top of the next knot, waits with
`s_waitcnt vmcnt(n)`, n = the number of STORE instructions the knot issued behind the prefetch (kMxGainStores = 4, + kMxCtgStores = 3 when every knot's
cost-to-go is written).  gfx9 retires vector memory operations in order, so the count is only right if the compiler emits exactly those stores and nothing else
between the prefetch and the wait -- a merged pair of stores (commit 9059827 had to keep two apart by hand), a spill, or an edit that adds a store turns the wait
into a race on the LDS operands with no diagnostic.  This test cross-compiles csrc/pddp_mx.hip for gfx950 (no GPU needed) and parses the knot loop of EVERY
prefetching instantiation; it also compiles the translation unit with a fifth gain store injected on purpose and requires the check to turn red."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mx_isa  # noqa: E402

# k_bp_mfma<FS, DIAGH, CAB, FUSE, HQQ>(Buffers<float>, ...): the float instantiations with the compact [A B] prefetch their operands
PREFETCHING = re.compile(r"^_ZN4pddp9k_bp_mfmaILb([01])ELb1ELb1ELb([01])ELb([01])EEEvNS_7BuffersIfEE")      # groups: FS, FUSE, HQQ


pytestmark = pytest.mark.skipif(not os.path.exists(mx_isa.HIPCC), reason="needs the ROCm compiler (cross-compiles for gfx950; no GPU needed)")


@pytest.fixture(scope="module")
def product_build():
    """ONE device compile of the product translation unit shared by the tests below: (assembly, resource remarks)"""
    return mx_isa.compile_asm(out="/tmp/isa/test_mx.s")


@pytest.fixture(scope="module")
def product_asm(product_build):
    return mx_isa.kernel_bodies(product_build[0])


def test_source_constants_match_the_checker():
    src = open(os.path.join(ROOT, "parallel-ddp_amd", "csrc", "bp_mfma.hpp")).read()
    m = re.search(r"constexpr int kMxGainStores = (\d+), kMxCtgStores = (\d+);", src)
    assert m and (int(m.group(1)), int(m.group(2))) == (mx_isa.K_GAIN, mx_isa.K_CTG)


def test_every_prefetching_instantiation_issues_exactly_the_counted_stores(product_asm):
    seen = 0
    for name, body in product_asm.items():
        m = PREFETCHING.match(name)
        if not m:
            continue
        seen += 1
        bad = mx_isa.check_prefetch_invariant(body, hqq=m.group(3) == "1", exact=not (m.group(1) == "1" and m.group(2) == "0"))
        assert not bad, f"{name}: " + "; ".join(bad)
    assert seen >= 5, f"only {seen} prefetching instantiations found: the name pattern no longer matches csrc/pddp_mx.hip"


def test_no_scratch_and_six_waves_for_the_benched_instantiation(product_build):
    # (the occupancy the timing of DESIGN.md section 4 was taken at: a spill inside the knot loop would also add vector memory operations the wait does not count)
    _, rem = product_build
    res = mx_isa.resources(rem)
    bench = next(v for k, v in res.items() if k.startswith("_ZN4pddp9k_bp_mfmaILb1ELb1ELb1ELb1ELb0EEEvNS_7BuffersIfEE"))
    assert bench["scratch"] == 0 and bench["occ"] >= 6, bench
    for k, v in res.items():
        if PREFETCHING.match(k):
            assert v["scratch"] == 0, (k, v)


def test_a_fifth_gain_store_turns_the_check_red():
    asm, _ = mx_isa.compile_asm(defs=["-DPDDP_MX_TEST_EXTRA_STORE=1"], out="/tmp/isa/test_mx_extra.s")
    bodies = mx_isa.kernel_bodies(asm)
    flagged = 0
    for name, body in bodies.items():
        m = PREFETCHING.match(name)
        if m:
            if m.group(1) == "1" and m.group(2) == "0":
                continue                                                  # (M > 1 without the fused maps: extra stores are allowed there)
            bad = mx_isa.check_prefetch_invariant(body, hqq=m.group(3) == "1")
            assert any("buffer stores per knot" in b for b in bad), (name, bad)
            flagged += 1
    assert flagged >= 3
