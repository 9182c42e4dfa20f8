"""CPU: what hipcc made of the LDS-DMA kernels (read from the build's own reports, svs_amd/lib/build/).

The phased GEMM (svs_amd/csrc/gemm_phased.h; the kernel of BASELINE.json configs[2] and configs[4],
i.e. a batch of the reference's np.dot, src/svs/kb.py:1623) keeps seven half-tiles of LDS-DMA in flight
behind counted `s_waitcnt vmcnt(N)`.  A spilled VGPR is reloaded with a scratch load followed by
`s_waitcnt vmcnt(0)`, which drains that ring: in round 2 the fp8 instantiation spilled three registers
and paid three drains per output tile.  These tests fail the build when that
comes back."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLD = os.path.join(ROOT, "svs_amd", "lib", "build")
RES = os.path.join(BLD, "svs_amd.resources.txt")
ISA = os.path.join(BLD, "svs_amd-hip-amdgcn-amd-amdhsa-gfx950.s")


@pytest.fixture(scope="module")
def build_reports():
    if not (os.path.exists(RES) and os.path.exists(ISA)):
        subprocess.run(["make", "-C", os.path.join(ROOT, "svs_amd", "csrc"), "-B", "-j2"], check=True)
    return RES, ISA


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return dict(zip(names, out.splitlines()))


def _resources(path):
    """{mangled kernel name: {field: int}} from -Rpass-analysis=kernel-resource-usage."""
    with open(path) as f:
        txt = f.read()
    table = {}
    for block in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = block.split()[0]
        fields = {}
        for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)")):
            m = re.search(pat, block)
            assert m, f"{name}: no '{key}' in the resource report"
            fields[key] = int(m.group(1))
        table[name] = fields
    return table


def test_no_fused_phased_kernel_uses_scratch(build_reports):
    table = _resources(build_reports[0])
    names = _demangle([n for n in table if "gemm_phased_kernel" in n])
    fused = {names[n]: table[n] for n in names if "gemm_phased_kernel<true" in names[n]}
    # f16 and fp8; 256- and 128-query tiles; default / nontemporal-corpus / A-B forms: all of them ship in the library
    assert len(fused) >= 10, sorted(fused)
    for want in ("<true, 1, 20, 256>", "<true, 2, 0, 256>", "<true, 1, 20, 128>", "<true, 2, 20, 128>"):
        assert any(want in k for k in fused), f"no gemm_phased_kernel{want} in the build"
    for k, r in fused.items():
        assert r["scratch"] == 0 and r["vgpr_spill"] == 0, f"{k}: {r} -- a spill reload drains the LDS-DMA ring"
        assert r["vgpr"] + r["agpr"] <= 256, f"{k}: {r} -- two waves per SIMD need <= 256 registers"
    plain = {names[n]: table[n] for n in names if "gemm_phased_kernel<false" in names[n]}
    for k, r in plain.items():
        assert r["scratch"] == 0, f"{k}: {r}"


def _function_body(isa_lines, mangled):
    start = next(i for i, l in enumerate(isa_lines) if l.startswith(mangled + ":"))
    end = next(i for i in range(start, len(isa_lines)) if isa_lines[i].startswith(".Lfunc_end"))
    return isa_lines[start:end]


def test_no_ring_drain_between_tile_loop_header_and_k_loop(build_reports):
    """Per output tile: side data staged, accumulators cleared, then the k loop.  Nothing in that
    stretch may wait for vmcnt(0) (the next tile's first half-tiles are already in flight), and the
    kernel may not touch scratch anywhere."""
    table = _resources(build_reports[0])
    names = _demangle([n for n in table if "gemm_phased_kernel" in n])
    with open(build_reports[1]) as f:
        isa = f.read().splitlines()
    checked = 0
    for mangled, pretty in names.items():
        if "gemm_phased_kernel<true" not in pretty:
            continue
        body = _function_body(isa, mangled)
        assert not [l for l in body if "scratch_" in l], f"{pretty}: scratch access"
        outer = next(i for i, l in enumerate(body) if "Loop Header: Depth=1" in l)
        inner = next(i for i in range(outer, len(body)) if "Inner Loop Header: Depth=2" in body[i])
        assert inner - outer < 600, f"{pretty}: tile-loop header and k loop are {inner - outer} lines apart (layout changed?)"
        stretch = body[outer:inner]
        assert any("buffer_load_dword" in l and " lds" in l for l in stretch), f"{pretty}: the side-data LDS-DMA is not where it was"
        drains = [l.strip() for l in stretch if re.search(r"s_waitcnt.*vmcnt\(0\)", l)]
        assert not drains, f"{pretty}: {len(drains)} ring-draining waits before the k loop"
        # the k loop itself: counted waits only
        # (the loop's blocks are the ones hipcc annotates with its header -- they need not be contiguous, the latch sits ABOVE
        #  the header, and wave-uniform branches inside the loop, the flush of the previous tile's candidates has some, end nothing)
        hdr_line = next(i for i in range(inner, max(inner - 6, 0), -1) if re.match(r"\.LBB\d+_\d+:", body[i]))
        hdr = re.match(r"\.L(BB\d+_\d+):", body[hdr_line]).group(1)
        mfma, loop_drains, cur_in = [], [], False
        for i, l in enumerate(body):
            if re.match(r"\.LBB\d+_\d+:", l):
                j, head = i + 1, l
                while j < len(body) and re.match(r"\s+;", body[j]):
                    head += body[j]
                    j += 1
                cur_in = i == hdr_line or f"Header={hdr} Depth=2" in head
            if cur_in and "v_mfma" in l:
                mfma.append(l)
            if cur_in and re.search(r"s_waitcnt.*vmcnt\(0\)", l):
                loop_drains.append(l.strip())
        assert len(mfma) >= 32, f"{pretty}: {len(mfma)} MFMAs in what should be the k loop (two k-tiles: 32 at fp8 128-query tiles .. 128 at f16 256-query tiles)"
        if re.search(r"gemm_phased_kernel<true, [12], (0|20), ", pretty):   # the shipped forms: nothing in the loop drains the LDS-DMA ring
            assert not loop_drains, f"{pretty}: {len(loop_drains)} vmcnt(0) waits inside the k loop"
        checked += 1
    assert checked >= 10


def test_no_shipped_gemm_kernel_touches_scratch_inside_its_k_loop(build_reports):
    """Every GEMM kernel the library dispatches to (gemm_phased / gemm_tiled / gemm_q16r, all instantiations): no VGPR spill
    and no scratch access -- with ONE documented exception, gemm_tiled_kernel<256, true, 1, 256> (the fp8 256 x 256 tile with
    the fused epilogue: the fallback for rows with an odd number of 128-byte k-tiles, d = 384, 640, 1152 ...), whose 128
    accumulators + epilogue temporaries do not fit 256 registers: its spills must all sit in the EPILOGUE, after the k loop,
    where nothing is in flight that a scratch reload's vmcnt(0) could drain (the scratch gate only
    looked at gemm_phased_kernel)."""
    table = _resources(build_reports[0])
    gemm = [n for n in table if any(k in n for k in ("gemm_phased_kernel", "gemm_tiled_kernel", "gemm_q16r_kernel", "gemm_f32_q16_kernel"))]
    names = _demangle(gemm)
    assert len(names) >= 40, len(names)
    with open(build_reports[1]) as f:
        isa = f.read().splitlines()
    allowed = "gemm_tiled_kernel<256, true, 1, 256>"
    seen_allowed = False
    for mangled, pretty in names.items():
        r = table[mangled]
        if allowed in pretty:
            seen_allowed = True
            body = _function_body(isa, mangled)
            mfma = [i for i, l in enumerate(body) if "v_mfma" in l]
            scr = [i for i, l in enumerate(body) if "scratch_" in l]
            assert len(mfma) >= 32 and scr, (len(mfma), len(scr))
            # the k loop is the only place with MFMAs; the epilogue follows it in the layout
            assert min(scr) > max(mfma), f"{pretty}: scratch access at line {min(scr)} of the kernel, before its last MFMA ({max(mfma)})"
            continue
        assert r["scratch"] == 0 and r["vgpr_spill"] == 0, f"{pretty}: {r}"
    assert seen_allowed
