"""csrc/tools/gcn_sched.py — the post-register-allocation scheduler the slab kernels (csrc/coflux_solver_slab.hip) are built
through: renaming of block-local values, list scheduling, recomputed s_waitcnt, and the symbolic verifier that compares
every emitted piece with the compiler's (same instructions on the same VALUES, every register the original writes ends
with the same value, LDS results used only behind a wait).  CPU only: text in, text out."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "climaocean.jl_amd", "csrc")
spec = importlib.util.spec_from_file_location("gcn_sched", os.path.join(CSRC, "tools", "gcn_sched.py"))
gs = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gs)

# two independent chains that share their temporaries (what register allocation leaves), an LDS read feeding one of them,
# a compare/select pair through vcc, a tied multiply-add
PIECE = """
	v_mul_f64 v[10:11], v[2:3], v[4:5]
	v_add_f64 v[10:11], v[10:11], v[6:7]
	v_mul_f64 v[20:21], v[10:11], v[10:11]
	v_lshl_add_u32 v30, v8, 3, 0
	ds_read_b128 v[12:15], v30 offset:1024
	v_mul_f64 v[10:11], v[6:7], v[6:7]
	v_add_f64 v[10:11], v[10:11], v[2:3]
	v_mul_f64 v[22:23], v[10:11], v[4:5]
	s_waitcnt lgkmcnt(0)
	v_fmac_f64_e32 v[20:21], v[12:13], v[14:15]
	v_cmp_lt_f64_e32 vcc, v[20:21], v[22:23]
	s_nop 1
	v_cndmask_b32_e32 v24, v20, v22, vcc
	v_mul_f64 v[10:11], v[20:21], v[22:23]
	v_rcp_f64_e32 v[26:27], v[10:11]
	v_fma_f64 v[28:29], v[26:27], v[10:11], -1.0
""".strip("\n").split("\n")


def _schedule(lines, pool=(100, 140)):
    insts = [gs.parse_inst(l) for l in lines]
    assert all(i is not None for i in insts)
    renamed = gs.rename_piece(insts, *pool)
    preds = gs.build_dag(insts)
    order = gs.list_schedule(insts, preds)
    return gs.emit(order, insts, preds, "\ts_waitcnt lgkmcnt(0)"), renamed, order


def test_rename_schedule_emit_verifies():
    out, renamed, order = _schedule(PIECE)
    assert renamed >= 2                      # the two earlier lives of v[10:11] at least
    assert order != list(range(len(order)))  # and the order moved
    gs.verify_piece(PIECE, out, (100, 140))
    text = "\n".join(out)
    assert "v[100:101]" in text or "v[102:103]" in text
    # what the last definition of a register leaves stays where the code behind the piece expects it
    assert any(l.strip().startswith("v_mul_f64 v[10:11], v[20:21], v[22:23]") for l in out)


@pytest.mark.parametrize("mutation", ["swap_dependent", "wrong_register", "drop_wait", "drop_instruction", "extra_write"])
def test_verifier_catches(mutation):
    out, _, _ = _schedule(PIECE)
    out = [l for l in out if l is not None]
    gs.verify_piece(PIECE, out, (100, 140))
    bad = list(out)
    code = lambda l: l.split(";")[0].strip()
    if mutation == "swap_dependent":
        i = next(k for k, l in enumerate(bad) if code(l).startswith("v_rcp_f64"))
        j = next(k for k, l in enumerate(bad) if code(l).startswith("v_fma_f64 v[28:29]"))
        bad[i], bad[j] = bad[j], bad[i]
    elif mutation == "wrong_register":
        i = next(k for k, l in enumerate(bad) if code(l).startswith("v_cndmask_b32"))
        bad[i] = bad[i].replace("v20", "v21")
    elif mutation == "drop_wait":
        bad = [l for l in bad if not code(l).startswith("s_waitcnt")]
    elif mutation == "drop_instruction":
        i = next(k for k, l in enumerate(bad) if code(l).startswith("v_mul_f64 v[22:23]") or ("v[22:23]" in code(l).split(",")[0]))
        del bad[i]
    elif mutation == "extra_write":
        bad.append("\tv_mov_b32_e32 v50, v2")
    with pytest.raises(SystemExit):
        gs.verify_piece(PIECE, bad, (100, 140))


def test_lgkmcnt_is_clamped_to_the_counter():
    many = ["\tv_lshl_add_u32 v30, v8, 3, 0"] + [f"\tds_read_b64 v[{40 + 2 * k}:{41 + 2 * k}], v30 offset:{8 * k}" for k in range(20)] + \
           ["\ts_waitcnt lgkmcnt(0)", "\tv_add_f64 v[10:11], v[40:41], v[78:79]"]
    out, _, _ = _schedule(many)
    import re
    assert all(int(m.group(1)) <= 15 for l in out if l for m in re.finditer(r"lgkmcnt\((\d+)\)", l))
    gs.verify_piece(many, out, (100, 140))


def test_built_slab_kernels_verify_piece_by_piece():
    """The build runs the verifier on every edited piece (a failure stops `make`); here the same over the build's own files,
    when they are there: every kernel of the slab translation unit, renamed from its own free registers."""
    dev = os.path.join(CSRC, "coflux_solver_slab.dev.s")
    if not os.path.exists(dev):
        pytest.skip("coflux_solver_slab.dev.s is a build artefact (python __graft_entry__.py build)")
    lines = open(dev).read().split("\n")
    fns = gs.matching_functions(lines, "ao_lean_line_kernel")
    assert len(fns) == 8      # COARE x (plain, fused, fused + tail, fused + tail + halo riders)
    total = 0
    for fn in fns:
        lo = gs.next_free_vgpr(lines, fn)
        lo += lo & 1
        _, edits = gs.process(lines, fn, "", False, pool=(lo, 255))      # verify_piece runs inside
        total += len(edits)
        assert any(e[5] >= 80 for e in edits), fn
    assert total > 100
