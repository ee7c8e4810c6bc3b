"""CPU: libqst.so loads without a GPU, exports every symbol the public headers declare, and its arena layout
agrees with the Python side. No compute entry point is called here."""
import ctypes
import ctypes as C
import os
import re

import numpy as np
import pytest

import quadruplet_sentence_transformer_amd  # noqa: F401
from quadruplet_sentence_transformer_amd import _lib
from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout, hf_param_views

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("include/qst.h", "include/qst_kernels.h"):
        src = open(os.path.join(ROOT, h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        names |= set(re.findall(r"\b(qst_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    decl = declared_symbols()
    assert len(decl) >= 30
    for name in sorted(decl):
        assert hasattr(lib, name), f"{name} declared in a header but not exported by libqst.so"
    # and the ctypes table binds exactly those
    assert set(_lib.SIGNATURES) == decl


def test_ctypes_structs_have_the_compiled_sizes():
    """The ctypes mirrors of the argument structs must match what the library was compiled with, field for field; sizes
    catch a forgotten or misplaced field (every struct ends in the fields added last)."""
    lib = _lib.load()
    for which, cls in enumerate([_lib.QstGemmArgs, _lib.QstLnEpi, _lib.QstFfnArgs, _lib.QstTnGroup, _lib.QstLnReduceBatch,
                                 _lib.QstDrop, _lib.QstAttnDesc]):
        assert lib.qst_abi_sizeof(which) == C.sizeof(cls), cls.__name__


def test_public_headers_are_self_contained_c(tmp_path):
    """include/*.h compile as plain C on their own (no HIP, no torch, nothing from csrc/)."""
    import subprocess
    for h in ("qst.h", "qst_kernels.h"):
        src = tmp_path / f"use_{h}.c"
        src.write_text(f'#include "{h}"\nint main(void) {{ return 0; }}\n')
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    assert "#include" not in open(os.path.join(ROOT, "include", "qst_kernels.h")).read().replace(
        "#include <stddef.h>", "").replace("#include <stdint.h>", "")


@pytest.mark.parametrize("name", sorted(PRESETS))
def test_arena_layout_python_equals_c(name):
    lib = _lib.load()
    cfg = PRESETS[name]
    c = _lib.make_config(cfg)
    segs, total = build_layout(cfg)
    assert lib.qst_arena_elems(c) == total
    assert lib.qst_arena_num_segments(c) == len(segs)
    for i, s in enumerate(segs):
        nm, off, n = ctypes.c_char_p(), ctypes.c_int64(), ctypes.c_int64()
        dec, gm = ctypes.c_int32(), ctypes.c_int32()
        assert lib.qst_arena_segment(c, i, nm, off, n, dec, gm) == 0
        assert (nm.value.decode(), off.value, n.value, bool(dec.value), bool(gm.value)) == \
               (s.name, s.offset, s.numel, s.decay, s.gemm)
    assert total % 256 == 0 and all(s.offset % 256 == 0 for s in segs)


def test_hf_views_cover_every_parameter_once():
    for name, cfg in PRESETS.items():
        segs, total = build_layout(cfg)
        so = {s.name: s for s in segs}
        cover = np.zeros(total, np.int32)
        for hf, seg, off, shape in hf_param_views(cfg):
            s = so[seg]
            cover[s.offset + off:s.offset + off + int(np.prod(shape))] += 1
        for s in segs:
            assert (cover[s.offset:s.offset + s.numel] == 1).all(), (name, s.name)
        assert cover.sum() == sum(s.numel for s in segs)


def test_decay_groups_follow_st_name_filter():
    # ST fit(): no decay for names containing 'bias', 'LayerNorm.bias', 'LayerNorm.weight'
    cfg = PRESETS["all-MiniLM-L6-v2"]
    so = {s.name: s for s in build_layout(cfg)[0]}
    for hf, seg, off, shape in hf_param_views(cfg):
        no_decay = any(nd in hf for nd in ["bias", "LayerNorm.bias", "LayerNorm.weight"])
        assert so[seg].decay == (not no_decay), hf


def test_param_count_matches_published_minilm():
    cfg = PRESETS["all-MiniLM-L6-v2"]
    assert sum(s.numel for s in build_layout(cfg)[0]) == 22565376      # trainable (no pooler)


def test_mpnet_bucket_host_function_matches_torch(golden_dir):
    lib = _lib.load()
    lut = np.load(os.path.join(golden_dir, "encoder_golden.npz"))["mpnet_bucket_lut"]
    got = np.array([lib.qst_rel_bucket_host(r, 32, 128) for r in range(-511, 512)])
    np.testing.assert_array_equal(got, lut)


def test_bad_config_is_rejected_without_a_device():
    lib = _lib.load()
    c = _lib.make_config(PRESETS["tiny-bert"])
    c.hidden_size = -1
    assert lib.qst_arena_elems(c) < 0
    assert lib.qst_strerror(-2).decode().startswith("unsupported")


def test_communicator_entry_points_check_their_arguments_without_a_gpu():
    """include/qst.h qst_comm_*: bad arguments are refused before RCCL is touched (no GPU, no RCCL needed)."""
    import ctypes as C
    from quadruplet_sentence_transformer_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    ident = C.create_string_buffer(128)
    assert lib.qst_comm_unique_id(None) == -1
    assert lib.qst_comm_init(0, 0, ident, C.byref(h)) == -1          # world <= 0
    assert lib.qst_comm_init(2, 2, ident, C.byref(h)) == -1          # rank outside [0, world)
    assert lib.qst_comm_init(0, 1, None, C.byref(h)) == -1
    assert lib.qst_allreduce_bucket(None, None, 4, 0, None) == -1
    assert lib.qst_comm_rank(None) == -1 and lib.qst_comm_world(None) == -1
    lib.qst_comm_destroy(None)                                       # no-op
    assert lib.qst_strerror(-6).decode().startswith("RCCL")
