"""CPU: what of the path can be pinned against the reference's own code: K1 whole (the function), K3 / K4 / K5 by their arithmetic
statements (second half of this file).

`change_volume<T>` (/root/reference/src/processor/audio-vol.cpp:75-100) needs only <algorithm> / <cstdint>, so `make -C oracle ref`
compiles it from where it lies into oracle/_ref/ (git-ignored; built by __graft_entry__.build() when /root/reference is present; the
.so files travel to the GPU box, the reference does not) — twice: libref_vol.so with g++ (xmake's default toolchain on the
reference's Linux target) and libref_vol_clang.so with clang.  Here: oracle/orc_nodes.c's K1 restatement ≡ BOTH builds, bit for bit,
wherever the C++ is defined (every float input; integer products inside the type's range), on the committed golden inputs and
outputs and on random frames of the sizes the node sees.  Integer products beyond the type's range are undefined behaviour in the
reference (audio-vol.cpp:98, no clamp): the g++ build wraps like `cvttss2si` + modular narrowing — what the oracle, the goldens and
the GPU kernel do — while clang's vectorised body SATURATES int16 (`packssdw`) and its scalar tail wraps; the last
test pins both facts, so "bit-exact int-PCM gain" is a statement about in-range products plus the g++ build's choice beyond them.

Every other loop of the path sits inside a process_payload body between FFmpeg / Boost calls: as FUNCTIONS they cannot be compiled
without stand-ins for headers this image lacks.  For K3 / K4 / K5 the arithmetic statements alone are compiled (below);
K6's two packed-integer conversions are one-line lambdas and are compiled as they stand (last test); K2 (pure data movement) and K6's
planar branches (their statements read AVFrame fields) stay pinned by the numpy restatement only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
REF_SRC = "/root/reference/src/processor/audio-vol.cpp"
BUILDS = {"g++": "libref_vol.so", "clang": "libref_vol_clang.so"}


def load_ref(which):
    so = os.path.join(REF_DIR, BUILDS[which])
    if not os.path.exists(so):
        if not os.path.exists(REF_SRC):
            pytest.skip("oracle/_ref was not built and /root/reference is not on this box")
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    L = C.CDLL(so)
    for n in ("ref_change_volume_f32", "ref_change_volume_s16", "ref_change_volume_s32"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        getattr(L, n).restype = None
    return L


@pytest.fixture(scope="module", params=["g++", "clang"])
def ref(request):
    L = load_ref(request.param)
    L.which = request.param
    return L


def in_range(x, v):
    """elements whose product the reference's C++ defines: float(x) * v, truncated, representable in x's type"""
    if x.dtype == np.float32:
        return np.ones(x.shape, bool)
    info = np.iinfo(x.dtype)
    p = np.trunc((x.astype(np.float32) * np.float32(v)).astype(np.float64))
    return (p >= info.min) & (p <= info.max)


def aligned(n, dtype, align=32):
    """the reference promises its compiler 32-byte aligned destination planes (audio-vol.cpp:95, av_frame_get_buffer(…, 32) at :173)"""
    raw = np.empty(n * np.dtype(dtype).itemsize + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n * np.dtype(dtype).itemsize].view(dtype)


def ref_change_volume(L, planes, volume):
    dt = planes[0].dtype
    fn = {np.dtype(np.float32): L.ref_change_volume_f32, np.dtype(np.int16): L.ref_change_volume_s16, np.dtype(np.int32): L.ref_change_volume_s32}[dt]
    src = [np.ascontiguousarray(p) for p in planes]
    dst = [aligned(p.size, dt) for p in src]
    sp = (C.c_void_p * len(src))(*[p.ctypes.data for p in src])
    dp = (C.c_void_p * len(dst))(*[p.ctypes.data for p in dst])
    fn(dp, sp, len(src), src[0].size, C.c_float(volume))
    return dst


def same_bits(a, b):
    return a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("key,vols", [("k1_f32", (0.0, 0.70710678, 1.0, 10.0)), ("k1_s16", (0.5, 0.70710678, 1.0, 3.0, 10.0)),
                                      ("k1_s32", (0.5, 0.70710678, 1.0, 3.0))])
def test_reference_code_reproduces_the_committed_golden(ref, golden, key, vols):
    """reference build == numpy golden == oracle on the committed vectors (tests/golden/nodes.npz); the g++ build also on their
    out-of-range products (k1_s16 at volume 3 and 10, k1_s32 at 3)"""
    g = golden["nodes"]
    x = g[key + "_in"]
    for v in vols:
        r = ref_change_volume(ref, [x], v)[0]
        o = orc.change_volume([x], v)[0]
        m = in_range(x, v) if ref.which == "clang" else np.ones(x.shape, bool)
        assert m.sum() > 500
        assert np.array_equal(r[m].view(np.uint8), o[m].view(np.uint8)), (key, v, "oracle differs from the reference's code")
        assert np.array_equal(r[m].view(np.uint8), g[f"{key}_v{v}"][m].view(np.uint8)), (key, v, "golden differs from the reference's code")


@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.int32])
@pytest.mark.parametrize("planes,elems", [(1, 2304), (2, 1152), (1, 8192), (2, 4096), (1, 1), (2, 7), (1, 33)])
def test_oracle_equals_reference_code_on_random_frames(ref, dtype, planes, elems):
    rng = np.random.default_rng(elems * 31 + planes)
    for v in (0.0, 0.25, 0.70710678, 1.0, 1.5, 9.999, 10.0):
        if dtype == np.float32:
            src = [rng.uniform(-1.5, 1.5, elems).astype(np.float32) for _ in range(planes)]
            src[0][:min(elems, 5)] = np.array([0.0, -0.0, 1e-40, np.inf, np.nan], np.float32)[:min(elems, 5)]
        else:
            # in range for every volume used here: the reference's float -> int conversion is only defined there
            lim = int(np.iinfo(dtype).max / 10.5)
            src = [rng.integers(-lim, lim, elems, dtype=np.int64).astype(dtype) for _ in range(planes)]
        r = ref_change_volume(ref, src, v)
        o = orc.change_volume(src, v)
        for a, b in zip(r, o):
            assert same_bits(a, b), (dtype, planes, elems, v)


@pytest.mark.parametrize("dtype", [np.int16, np.int32])
def test_out_of_range_integer_products(dtype):
    """volume * sample beyond the integer type (full-range samples up to the node's maximum volume, config.hpp:58: 10) is undefined
    in the reference's C++.  g++ -O3: `cvttss2si` (0x80000000 on overflow), then modular narrowing — the oracle's choice, bit for bit.
    clang -O3: the same for int32; for int16 its vector body saturates and its scalar tail wraps, so it agrees with the oracle on
    in-range products only — which is all the reference defines."""
    gxx, clang = load_ref("g++"), load_ref("clang")
    info = np.iinfo(dtype)
    rng = np.random.default_rng(7)
    x = rng.integers(info.min, info.max, 4099, dtype=np.int64, endpoint=True).astype(dtype)
    x[:4] = [info.min, info.max, info.min + 1, info.max - 1]
    for v in (1.0000001, 2.0, 3.0, 10.0):
        o = orc.change_volume([x], v)[0]
        r = ref_change_volume(gxx, [x], v)[0]
        assert same_bits(r, o), (dtype, v, int(np.count_nonzero(r != o)))
        c = ref_change_volume(clang, [x], v)[0]
        m = in_range(x, v)
        assert 0 < m.sum() and (v < 2.0 or m.sum() < x.size)
        assert np.array_equal(c[m], o[m]), (dtype, v)
        if dtype == np.int16 and v >= 2.0:
            assert not np.array_equal(c, o)                      # compilers disagree where the reference leaves the result undefined


# ---------------------------------------------------------------------------------------------- K3 / K4 / K5: arithmetic statements
# oracle/ref_mix_tu.sh streams the arithmetic STATEMENTS of three more loops from the reference into wrapper loops (the loop headers
# and the variable declarations are the builder's, repeating the reference's types; no FFmpeg / Boost header is imitated):
#   K3 audio-amix.cpp:298-306, K4 audio-bimix.cpp:310-311,315-316, K5 audio-bimix.cpp:627.
# So for K3-K5 the pin covers what the reference computes per sample and in which order, not the frame plumbing around it.
MIX_BUILDS = {"g++": "libref_mix.so", "clang": "libref_mix_clang.so"}


@pytest.fixture(scope="module", params=["g++", "clang"])
def refmix(request):
    so = os.path.join(REF_DIR, MIX_BUILDS[request.param])
    if not os.path.exists(so):
        if not os.path.exists(REF_SRC):
            pytest.skip("oracle/_ref was not built and /root/reference is not on this box")
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    L = C.CDLL(so)
    L.ref_amix_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.ref_bimix_f32.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p, C.c_size_t]
    L.ref_bimix2_downmix_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    for f in (L.ref_amix_f32, L.ref_bimix_f32, L.ref_bimix2_downmix_f32):
        f.restype = None
    return L


def special(n, seed):
    """uniform noise with the awkward values in front: zeros of both signs, subnormals, huge, Inf, NaN"""
    x = orc.fill_uniform(n, seed).copy()
    edge = np.array([0.0, -0.0, 1e-40, -1e-40, 1.1754944e-38, 3.4e38, -3.4e38, np.inf, -np.inf, np.nan, 1.0, -1.0], np.float32)
    x[:min(n, edge.size)] = edge[:min(n, edge.size)]
    return x


@pytest.mark.parametrize("n_in", [1, 2, 3, 16])
def test_k3_mix_statements_of_the_reference(refmix, n_in):
    S = 4099
    rng = np.random.default_rng(n_in)
    inL = [special(S, 10 * n_in + i) if i == 0 else orc.fill_uniform(S, 10 * n_in + i) for i in range(n_in)]
    inR = [orc.fill_uniform(S, 100 + 10 * n_in + i) for i in range(n_in)]
    vol = rng.uniform(0, 1, n_in).astype(np.float32)
    oL, oR = orc.amix(inL, inR, vol)
    # datas[i] = {left plane, right plane} as uint8_t** (what swr_convert fills, audio-amix.cpp:251-269)
    pairs = [(C.c_void_p * 2)(inL[i].ctypes.data, inR[i].ctypes.data) for i in range(n_in)]
    datas = (C.c_void_p * n_in)(*[C.addressof(p) for p in pairs])
    rL, rR = np.empty(S, np.float32), np.empty(S, np.float32)
    refmix.ref_amix_f32(datas, vol.ctypes.data, n_in, rL.ctypes.data, rR.ctypes.data, S)
    assert same_bits(rL, oL) and same_bits(rR, oR)


@pytest.mark.parametrize("bias", [-1.0, -0.3, 0.0, 0.25, 1.0])
def test_k4_bimix_statements_of_the_reference(refmix, bias):
    S = 4099
    a = [special(S, 21), orc.fill_uniform(S, 22), orc.fill_uniform(S, 23), special(S, 24)]
    oL, oR = orc.bimix(*a, bias)
    rL, rR = np.empty(S, np.float32), np.empty(S, np.float32)
    refmix.ref_bimix_f32(*[x.ctypes.data for x in a], C.c_float(bias), rL.ctypes.data, rR.ctypes.data, S)
    assert same_bits(rL, oL) and same_bits(rR, oR)


def test_k5_downmix_statement_of_the_reference(refmix):
    S = 4099
    l, r = special(S, 31), special(S, 32)[::-1].copy()
    m = orc.bimix2_downmix(l, r)
    ref = np.empty(S, np.float32)
    refmix.ref_bimix2_downmix_f32(l.ctypes.data, r.ctypes.data, ref.ctypes.data, S)
    assert same_bits(ref, m)          # incl. the subnormal halving: (l + r) * 0.5 is a DOUBLE product rounded once


@pytest.mark.parametrize("earlier_channel", [0, 1])
def test_k5_interleave_loops_of_the_reference(refmix, earlier_channel):
    """audio-bimix.cpp:799-803 (single-sided frame) and :836-850 (unaligned + aligned frame), loops and headers as they stand"""
    refmix.ref_bimix2_interleave_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    refmix.ref_bimix2_interleave_f32.restype = None
    e, l = special(3000, 41), special(2500, 42)
    for unaligned, aligned in ((0, 2500), (123, 2500), (500, 2000), (2999, 1), (3000, 0), (0, 0)):
        want = orc.bimix2_interleave(e, l, unaligned, aligned, earlier_channel)
        got = np.empty(2 * (unaligned + aligned), np.float32)
        refmix.ref_bimix2_interleave_f32(got.ctypes.data, e.ctypes.data, e.size, l.ctypes.data, l.size, unaligned, aligned, earlier_channel)
        assert same_bits(got, want), (unaligned, aligned)
    want = orc.bimix2_interleave(e, None, e.size, 0, earlier_channel)
    got = np.empty(2 * e.size, np.float32)
    refmix.ref_bimix2_interleave_f32(got.ctypes.data, e.ctypes.data, e.size, None, 0, 0, 0, earlier_channel)
    assert same_bits(got, want)


@pytest.mark.parametrize("dtype,fn,fmt", [(np.int16, "ref_k6_s16_packed", orc.FMT_S16), (np.int32, "ref_k6_s32_packed", orc.FMT_S32)])
def test_k6_packed_integer_lambdas_of_the_reference(refmix, dtype, fn, fmt):
    """audio-velocity.cpp:186 (S16 / 32768.0f) and :207 (S32 / 2147483648.0f): the lambdas as they stand, over every int16 value and
    over random + extreme int32 values"""
    f = getattr(refmix, fn)
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    f.restype = None
    info = np.iinfo(dtype)
    if dtype == np.int16:
        x = np.arange(info.min, info.max + 1, dtype=np.int64).astype(dtype)
    else:
        rng = np.random.default_rng(6)
        x = rng.integers(info.min, info.max, 1 << 16, dtype=np.int64, endpoint=True).astype(dtype)
        x[:6] = [info.min, info.max, 0, -1, 1, info.min + 1]
    if x.size % 2:
        x = x[:-1]
    got = np.empty(x.size, np.float32)
    f(x.ctypes.data, got.ctypes.data, x.size)
    rc, want = orc.to_f32_interleaved(fmt, [x], x.size // 2, 2)      # packed stereo: one plane of S * 2 samples
    assert rc == 0 and same_bits(got, want)


def test_k3_weight_renormalisation_of_the_reference(refmix):
    """audio-amix.cpp:379-387: the unlocked weights are scaled to sum 1 (floor 0.001); loops with their headers"""
    refmix.ref_amix_normalise.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    refmix.ref_amix_normalise.restype = None
    rng = np.random.default_rng(9)
    for n in (1, 2, 5, 16):
        for trial in range(20):
            v = rng.uniform(0, 1, n).astype(np.float32)
            locks = (rng.uniform(0, 1, n) < 0.3).astype(np.uint8)
            if trial == 0:
                v[:] = 0                                  # the 0.001 floor
            if trial == 1:
                locks[:] = 1                              # nothing unlocked
            want = orc.amix_normalise(v, locks)
            got = v.copy()
            refmix.ref_amix_normalise(got.ctypes.data, locks.ctypes.data, n)
            assert same_bits(got, want), (n, trial)


# ------------------------------------------------------------------------------------------------------------------------------------
# The boundary's TEXT: the plugin ABC's pure-virtual list and every processor's identifier, display name, singleton flag,
# pins and JSON keys, read out of the reference's sources as text (build container only) and compared with what the C++ host mirror's
# registry holds (`tests/host/selftest registry` prints it from default-constructed nodes).  A renamed pin, a missing override, a dropped
# JSON key or another identifier fails here.
REF_ROOT = "/root/reference"
HOST_DIR = os.path.join(ROOT, "nodey-audio-editor_amd", "host")
REF_CLASSES = {      # class -> (source file, header) of the reference
    "Audio_vol": ("src/processor/audio-vol.cpp", "include/processor/audio-vol.hpp"),
    "Audio_amix": ("src/processor/audio-amix.cpp", "include/processor/audio-amix.hpp"),
    "Audio_bimix": ("src/processor/audio-bimix.cpp", "include/processor/audio-bimix.hpp"),
    "Audio_bimix_v2": ("src/processor/audio-bimix.cpp", "include/processor/audio-bimix.hpp"),
    "Velocity_modifier": ("src/processor/audio-velocity.cpp", "include/processor/audio-velocity.hpp"),
    "Pitch_modifier": ("src/processor/audio-velocity.cpp", "include/processor/audio-velocity.hpp"),
}


def _need_reference():
    if not os.path.exists(os.path.join(REF_ROOT, "include", "infra", "processor.hpp")):
        pytest.skip("/root/reference is not on this box (the boundary text is pinned in the build container)")


def _strip_comments(text):
    import re
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def _pure_virtuals(header_text, class_name="Processor"):
    """names of the `virtual ... name(...) [const] = 0;` members of one class"""
    import re
    text = _strip_comments(header_text)
    start = re.search(r"\bclass\s+%s\b[^;{]*\{" % class_name, text).end()
    depth, i = 1, start
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    body = text[start:i]
    # nested classes / structs of the ABC (Product, Info, Runtime_error ...) have no pure virtuals of their own in either tree
    return sorted(set(re.findall(r"virtual\s+[^;{}]*?\b(\w+)\s*\([^;{}]*\)\s*(?:const\s*)?=\s*0\s*;", body)))


def _function_body(text, qualified_name):
    """text between the braces of `qualified_name(...) [const] {` (first definition)"""
    import re
    m = re.search(re.escape(qualified_name) + r"\s*\([^)]*\)\s*(?:const\s*)?\{", text)
    if not m:
        return None
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1]


def _pattern(s):
    """input_1 / volumes0 -> input_{} / volumes{}: one name per std::format pattern"""
    import re
    return re.sub(r"\d+", "{}", s)


def _dedupe(seq):
    out = []
    for x in seq:
        if x not in out:
            out.append(x)
    return out


def _reference_boundary(cls):
    import re
    src_path, hdr_path = REF_CLASSES[cls]
    src = _strip_comments(open(os.path.join(REF_ROOT, src_path), encoding="utf-8").read())
    hdr = _strip_comments(open(os.path.join(REF_ROOT, hdr_path), encoding="utf-8").read())
    info = _function_body(src, f"{cls}::get_processor_info")
    ident = re.search(r"\.identifier\s*=\s*\"([^\"]*)\"", info).group(1)
    name = re.search(r"\.display_name\s*=\s*\"([^\"]*)\"", info).group(1)
    singleton = re.search(r"\.singleton\s*=\s*(true|false)", info).group(1) == "true"
    pins_body = _function_body(src, f"{cls}::get_pin_attributes")
    ids = re.findall(r"\.identifier\s*=\s*(?:std::format\(\s*)?\"([^\"]*)\"", pins_body)
    dirs = re.findall(r"\.is_input\s*=\s*(true|false)", pins_body)
    types = re.findall(r"\.type\s*=\s*typeid\((\w+)\)", pins_body)
    assert len(ids) == len(dirs) == len(types) and ids, (cls, ids, dirs, types)
    ser = _function_body(src, f"{cls}::serialize")
    if ser is None:
        # defined in the class body: `virtual Json::Value serialize() const { return {}; }`
        m = re.search(r"class\s+%s\b.*?serialize\s*\(\s*\)\s*const\s*\{([^}]*)\}" % cls, hdr, re.S)
        assert m, f"{cls}::serialize not found in the reference"
        ser = m.group(1)
    keys = re.findall(r"value\[\s*(?:std::format\(\s*)?\"([^\"]*)\"", ser)
    return {"identifier": ident, "display_name": name, "singleton": singleton,
            "pins": [(_pattern(i), d == "true", t) for i, d, t in zip(ids, dirs, types)], "json_keys": sorted(set(_pattern(k) for k in keys))}


def _mirror_registry():
    import json
    exe = os.path.join(ROOT, "tests", "host", "selftest")
    if not os.path.exists(exe):
        for d in (HOST_DIR, os.path.join(ROOT, "tests", "host")):
            r = subprocess.run(["make", "-C", d, "-j4"], capture_output=True, text=True)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([exe, "registry"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    procs = [json.loads(line[5:]) for line in r.stdout.splitlines() if line.startswith("PROC ")]
    return {p["identifier"]: p for p in procs}


def test_boundary_pure_virtuals_equal_the_reference_abc():
    """infra::Processor of the mirror declares exactly the reference's pure virtuals (include/infra/processor.hpp:86-113) — draw_title and
    draw_content among them — and `selftest registry` instantiates every registered class through Info::generate, which compiles only if
    the class overrides all of them: the adapters are concrete against the real ABC."""
    _need_reference()
    ref = _pure_virtuals(open(os.path.join(REF_ROOT, "include", "infra", "processor.hpp"), encoding="utf-8").read())
    mine = _pure_virtuals(open(os.path.join(HOST_DIR, "infra", "processor.hpp"), encoding="utf-8").read())
    assert ref == ["deserialize", "draw_content", "draw_title", "get_pin_attributes", "get_processor_info_non_static", "process_payload", "serialize"], ref
    assert mine == ref
    # every adapter class DECLARES each of them (a class that inherited an empty body would pass the compiler but not the real tree)
    import re
    for hdr in ("audio-vol.hpp", "audio-mix.hpp", "audio-velocity.hpp"):
        text = _strip_comments(open(os.path.join(HOST_DIR, "processor", hdr), encoding="utf-8").read())
        for m in re.finditer(r"class\s+(\w+)\s*:\s*public\s+infra::Processor\s*\{", text):
            depth, i = 1, m.end()
            while depth:
                depth += {"{": 1, "}": -1}.get(text[i], 0)
                i += 1
            body = text[m.end():i]
            for fn in ref:
                assert re.search(r"\b%s\s*\(" % fn, body), f"{m.group(1)} does not declare {fn}"
    assert len(_mirror_registry()) == 7


@pytest.mark.parametrize("cls", sorted(REF_CLASSES))
def test_boundary_text_of_every_processor_equals_the_reference(cls):
    """identifier, display name, singleton flag, pin identifiers / directions / product type and serialize() keys of one processor class:
    the reference's source text against the mirror's registry (SURVEY.md §8b "Identifiers / pins / JSON keys")."""
    _need_reference()
    ref = _reference_boundary(cls)
    reg = _mirror_registry()
    assert ref["identifier"] in reg, f"{cls}: identifier {ref['identifier']} is not registered by the mirror"
    got = reg[ref["identifier"]]
    assert got["display_name"] == ref["display_name"]
    assert got["singleton"] == ref["singleton"] and got["same_info_non_static"]
    assert all(t == "Audio_stream" for _, _, t in ref["pins"])
    assert all(is_stream for _, _, is_stream in got["pins"]), "every pin carries an Audio_stream and generates one"
    assert _dedupe([(_pattern(i), d) for i, d, _ in got["pins"]]) == _dedupe([(i, d) for i, d, _ in ref["pins"]])
    assert sorted(set(_pattern(k) for k in got["json_keys"])) == ref["json_keys"]


def test_boundary_register_list_covers_the_reference_hot_path_nodes():
    """src/register.cpp:16-23 registers eight classes; the six on the hot path are replaced under their identifiers, audio_input / audio_output
    (codecs, device I/O: out of scope) stay the reference's, audio_spectrum is new"""
    _need_reference()
    import re
    text = _strip_comments(open(os.path.join(REF_ROOT, "src", "register.cpp"), encoding="utf-8").read())
    registered = re.findall(r"register_processor<\s*(?:processor::)?(\w+)\s*>", text)
    assert sorted(registered) == sorted(list(REF_CLASSES) + ["Audio_input", "Audio_output"]), registered
    reg = _mirror_registry()
    assert sorted(reg) == sorted([_reference_boundary(c)["identifier"] for c in REF_CLASSES] + ["audio_spectrum"])
