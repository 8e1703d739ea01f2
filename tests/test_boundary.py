"""CPU: the C-ABI library loads, exports every symbol include/nsx.h declares, and the product never
touches the oracle (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    syms = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            txt = open(os.path.join(inc, f)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            syms |= set(re.findall(r"\b(nsx_[a-z0-9_]+)\s*\(", txt))
    return syms


def test_library_exports_every_declared_symbol():
    from nersemble_amd import _lib
    handle = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 9
    for s in declared:
        assert hasattr(handle, s), f"libnsx.so does not export {s}"
    # the ctypes signature table covers exactly the header
    assert set(_lib.SIGNATURES) == declared
    assert handle.nsx_version() >= 120


def test_no_exported_path_reads_the_environment():
    """Rounds 2-4 selected kernel variants -- some of them timing probes with wrong results on purpose -- through NSX_*
    environment variables read inside entry points of the C ABI.  Round 5: launch shapes are explicit options
    (nsx_set_option), the variants are gone; neither the sources nor the built library reference getenv."""
    src_dir = os.path.join(ROOT, "nersemble_amd", "csrc")
    for f in sorted(os.listdir(src_dir)):
        if f.endswith((".hip", ".h", ".cpp")):
            txt = open(os.path.join(src_dir, f)).read()
            assert not re.search(r"\b(secure_)?getenv\b|\benviron\b", txt), f"{f} reads the environment"
    from nersemble_amd import _lib
    blob = open(_lib.SO_PATH, "rb").read()
    assert b"getenv" not in blob, "libnsx.so imports getenv"
    for name in (b"NSX_DEFORM_FWD", b"NSX_HASHGRID_FWD", b"NSX_ADAM_BLOCKS_PER_CU", b"NSX_MLP_BWD"):
        assert name not in blob


def test_launch_shape_options_are_explicit_and_checked():
    from nersemble_amd import _lib
    L = _lib.lib()
    defaults = {_lib.NSX_OPT_ADAM_BLOCKS_PER_CU: 5, _lib.NSX_OPT_MLP_BWD_HALF_BLOCKS_PER_CU: 2,
                _lib.NSX_OPT_MLP_BWD0_HALF_BLOCKS_PER_CU: 2}
    for opt, dflt in defaults.items():
        assert L.nsx_get_option(opt) == dflt
        assert L.nsx_set_option(opt, 4) == 0 and L.nsx_get_option(opt) == 4
        assert L.nsx_set_option(opt, 0) != 0 and b"takes" in L.nsx_last_error()       # out of range: refused, value kept
        assert L.nsx_set_option(opt, 9) != 0 and L.nsx_get_option(opt) == 4
        assert L.nsx_set_option(opt, dflt) == 0
    opt = _lib.NSX_OPT_LP_ONE_LAUNCH                       # a switch: 0 / 1, on by default
    assert L.nsx_get_option(opt) == 1
    assert L.nsx_set_option(opt, 0) == 0 and L.nsx_get_option(opt) == 0
    assert L.nsx_set_option(opt, 2) != 0 and L.nsx_set_option(opt, -1) != 0 and L.nsx_get_option(opt) == 0
    assert L.nsx_set_option(opt, 1) == 0
    assert L.nsx_set_option(99, 1) != 0 and L.nsx_get_option(99) < 0


def test_library_collectives_refuse_bad_arguments_before_touching_rccl():
    """csrc/comm.hip: argument errors are reported through the C ABI's error state, without a GPU and without a communicator."""
    from nersemble_amd import _lib
    L = _lib.lib()
    ident = (ctypes.c_uint8 * _lib.NSX_COMM_ID_BYTES)()
    comm = ctypes.c_void_p()
    assert L.nsx_comm_create(ident, 2, 5, ctypes.byref(comm)) != 0 and b"rank 5 of 2" in L.nsx_last_error()
    assert L.nsx_comm_create(None, 1, 0, ctypes.byref(comm)) != 0 and b"NULL" in L.nsx_last_error()
    assert comm.value is None
    assert L.nsx_comm_destroy(None) == 0                       # destroying nothing is fine
    assert L.nsx_comm_world_size(None) < 0 and L.nsx_comm_rank(None) < 0
    assert L.nsx_comm_all_reduce_sum(None, None, 4, None) != 0 and b"nsx_comm_all_reduce_sum" in L.nsx_last_error()
    lay = _lib.step_struct("nsx_lp_layout")()
    assert L.nsx_lp_layout_make(4, 1000, 8, 32, 8, ctypes.byref(lay)) == 0
    g = _lib.GridGeom()
    args_f = [ctypes.byref(lay), None, -1] + [None, None, 0, None, None, 0, 1] + [None] * 5 + [ctypes.byref(g)] + [None] * 6
    assert L.nsx_lp_forward(*args_f) != 0 and b"NULL communicator" in L.nsx_last_error()
    args_b = [ctypes.byref(lay), None, -1] + [None, None, None, 0] + [None] * 7 + [ctypes.byref(g)] + [None] * 7 + [1, None, None, None]
    assert L.nsx_lp_backward(*args_b) != 0 and b"NULL communicator" in L.nsx_last_error()


def test_bucket_tail_calls_check_their_arguments():
    from nersemble_amd import _lib
    L = _lib.lib()
    sizes = (ctypes.c_int64 * 17)(*([1] * 17))
    ptrs = (ctypes.c_void_p * 17)()
    assert L.nsx_bucket_pack(None, ptrs, sizes, 17, None) != 0 and b"17 pieces (limit 16)" in L.nsx_last_error()
    assert L.nsx_bucket_pack(None, ptrs, sizes, 2, None) != 0 and b"piece 0" in L.nsx_last_error()        # NULL piece of size 1
    assert L.nsx_bucket_pack(None, None, None, 0, None) == 0                                              # nothing to do
    assert L.nsx_bucket_unpack(None, 5, 0.5, None, None, None, 0, None) != 0 and b"NULL buffer" in L.nsx_last_error()
    assert L.nsx_bucket_unpack(None, 0, 0.5, None, None, None, 0, None) == 0


def test_error_reporting_across_abi():
    from nersemble_amd import _lib
    g = _lib.GridGeom()
    rc = _lib.lib().nsx_grid_geometry(99, 1.5, 16, 19, ctypes.byref(g))
    assert rc != 0
    assert b"n_levels" in _lib.lib().nsx_last_error()
    with pytest.raises(RuntimeError, match="n_levels"):
        _lib.check(rc, "nsx_grid_geometry")


def test_native_geometry_equals_oracle_geometry():
    import oracle
    from nersemble_amd import _lib
    from tests.helpers import REF_GEOM_KW, SMALL_GEOM_KW
    for kw in (REF_GEOM_KW, SMALL_GEOM_KW, dict(n_levels=8, per_level_scale=2.0, base_resolution=4, log2_hashmap_size=10)):
        a, b = _lib.grid_geometry(**kw), oracle.grid_geometry(**kw)
        for l in range(kw["n_levels"]):
            assert a.scale[l] == b.scale[l] and a.res[l] == b.res[l]
            assert a.size[l] == b.size[l] and a.offset[l] == b.offset[l]
        assert a.total_entries == b.total_entries


def test_product_never_imports_oracle_or_reference():
    pkg = os.path.join(ROOT, "nersemble_amd")
    bad = []
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "nsx_oracle" in txt \
                        or "/root/reference" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, f"product code references the oracle/reference: {bad}"


def test_native_ops_refuse_cpu_tensors():
    import torch
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    he = HashEnsemble(HashEnsembleConfig(2, TCNNHashEncodingConfig(n_levels=2, log2_hashmap_size=8)))
    with pytest.raises(RuntimeError, match="no CPU fallback|device tensors"):
        he(torch.rand(4, 3), torch.rand(4, 2))


def test_hash_ensemble_state_dict_uses_reference_keys():
    import numpy as np
    import torch
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    cfg = HashEnsembleConfig(16, TCNNHashEncodingConfig(n_levels=3, log2_hashmap_size=9))
    he = HashEnsemble(cfg)
    sd = he.state_dict()
    assert sorted(sd) == [f"hash_encodings.{c}.params" for c in range(4)]
    total = he.geom.total_entries
    assert all(v.shape == (total * 8,) for v in sd.values())
    # tcnn feature j = p*2+f of encoding c is logical grid h = c*4+p  (hash_ensemble.py:107-112)
    tc = torch.stack([sd[f"hash_encodings.{c}.params"].reshape(total, 8) for c in range(4)])
    for h in (0, 5, 15):
        c, p = divmod(h, 4)
        for f in (0, 1):
            assert torch.equal(tc[c, :, p * 2 + f], he.tables[:, f, h])
    he2 = HashEnsemble(cfg, seed=99)
    he2.load_state_dict(sd)
    assert torch.equal(he2.tables, he.tables)


@pytest.mark.parametrize("kind,name", [(0, "nsx_step_sample"), (1, "nsx_step_main")])
def test_step_driver_structs_have_the_compilers_layout(kind, name):
    """The training-step drivers take their arguments as structs (include/nsx.h).  The binding's ctypes mirrors are built
    from the header text; here they are held to the COMPILED layout: same size, and every field written through ctypes
    arrives in the library where the header says (``nsx_step_echo`` reads each field by name on the C side)."""
    from nersemble_amd import _lib
    cls = _lib.step_struct(name)
    L = _lib.lib()
    assert ctypes.sizeof(cls) == L.nsx_step_sizeof(kind)
    assert ctypes.sizeof(_lib.step_struct("nsx_step_plan")) == L.nsx_step_sizeof(2)
    s = cls()
    want = []
    v = 1000
    for fname, ftype in cls._fields_:
        if hasattr(ftype, "_length_"):
            arr = getattr(s, fname)
            for i in range(ftype._length_):
                arr[i] = float(v)
                want.append(float(v))
                v += 1
        elif ftype is ctypes.c_float:
            setattr(s, fname, float(v) + 0.5)
            want.append(float(v) + 0.5)
            v += 1
        else:                                   # pointers and integers
            setattr(s, fname, v)
            want.append(float(v))
            v += 1
    out = (ctypes.c_double * 256)()
    n = L.nsx_step_echo(kind, ctypes.byref(s), out, 256)
    assert n == len(want), (n, len(want))
    assert list(out[:n]) == want


def test_step_plan_offsets_are_disjoint_and_aligned():
    from nersemble_amd import _lib
    plan = _lib.step_struct("nsx_step_plan")()
    _lib.check(_lib.lib().nsx_step_plan_make(100_003, 4096, 24, 32, 0, 1, ctypes.byref(plan)), "nsx_step_plan_make")
    groups = {"sample": ("m_ri", "k_total", "sample_bytes"), "fwd": ("f_pos", "f_per_ray", "fwd_bytes"),
              "grad": ("g_head", "g_code_hash", "grad_bytes"), "bwd": ("b_grgb", "b_deform", "bwd_bytes")}
    names = [f for f, _ in plan._fields_]
    for first, last, total in groups.values():
        offs = [getattr(plan, n) for n in names[names.index(first):names.index(last) + 1]]
        offs = [o for o in offs]
        assert offs[0] == 0 and all(o % 256 == 0 for o in offs)
        assert all(b >= a for a, b in zip(offs, offs[1:])) and getattr(plan, total) >= offs[-1]
    assert plan.S == 100_003 and plan.R == 4096 and plan.g_zero_end == plan.g_code_hash
    assert plan.b_zero_end - plan.b_ds >= 100_003 * (4 + 6 + 32)
    assert _lib.lib().nsx_step_plan_make(0, 4096, 24, 32, 0, 1, ctypes.byref(plan)) != 0
