"""CPU-only checks of the drop-in boundary: libbasic_dsp_hip.so loads without a GPU and exports
every symbol include/basic_dsp_hip.h declares (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "basic_dsp_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)  # preprocessor lines (#define X (-1) ...)
    text = re.sub(r"typedef[^;{]*\(\s*\*[^;]*;", "", text)  # function-pointer typedefs are not exports
    # declarations bound to another symbol name (powf32 & co. clash with <math.h>): the label is what is exported
    text = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*(\s*\([^;{}()]*\))\s*BDSP_FACADE_SYMBOL\((\w+)\)\s*;", r"\2\1;", text)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined", "void")))  # "void (*fn)(...)" members


def test_header_declares_the_expected_surface():
    names = declared_functions()
    # B1: the five GpuSupport<T> functions x two precisions (vector/src/gpu_support/mod.rs:18-46)
    for base in ("has_gpu_support", "is_supported_fft_len", "fft", "convolve_vector", "overlap_discard"):
        for sfx in ("f32", "f64"):
            assert "bdsp_hip_%s_%s" % (base, sfx) in names
    # B2: facade names (interop/src/facade32.rs) for the hot path
    for base in ("new", "delete_vector", "plain_fft", "fft", "windowed_fft", "plain_ifft", "ifft",
                 "magnitude", "convolve_signal", "interpolatef", "real_scale", "real_offset",
                 "complex_scale", "multiply_complex_exponential", "conj", "mul", "swap_halves",
                 "zero_pad", "overwrite_data", "data", "get_len"):
        for sfx in ("32", "64"):
            assert base + sfx in names, base + sfx
    assert len(names) > 110


def test_library_loads_and_exports_every_declared_symbol():
    import basic_dsp_amd._lib as L
    assert os.path.exists(L.LIB_PATH)
    lib = C.CDLL(L.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_no_gpu_reporting():
    import basic_dsp_amd as b
    assert b.lib.bdsp_hip_version().startswith(b"basic_dsp_hip")
    # without a device the probe answers 0 and explains why; with one it answers 1
    ok = b.lib.bdsp_hip_has_gpu_support_f32()
    assert ok in (0, 1)
    if not ok:
        assert b.last_error() != ""
        with pytest.raises(b.BackendError):
            b.DspVec([1.0, 2.0])  # the product path fails loudly, it never falls back to the CPU


def test_is_supported_fft_len_contract():
    """GpuSupport::is_supported_fft_len with the B1 size policy (round 6): real input refused like ocl/mod.rs:277-281; complex
    lengths answered 0 below the measured crossover with the caller's rustfft (the trait's own way to decline,
    time_freq/mod.rs:41-44), 1 above it for ANY length; the thresholds are readable and settable (0 = never decline)."""
    import basic_dsp_amd as b
    L = b._lib
    f32, f64 = b.lib.bdsp_hip_is_supported_fft_len_f32, b.lib.bdsp_hip_is_supported_fft_len_f64
    get, put = b.lib.bdsp_hip_b1_policy_get, b.lib.bdsp_hip_b1_policy_set
    defaults = [get(k) for k in range(4)]
    try:
        # the shipped defaults: profiles/r06_b1_crossover.txt (f32 8192 points, f64 16384 points; 65 536 / 98 304 points x taps)
        assert defaults == [2 * 8192, 2 * 16384, 65536, 98304]
        assert f32(0, 1 << 20) == 0 and f64(0, 1 << 20) == 0                 # real input
        assert f32(1, 2 * 8192) == 1 and f32(1, 2 * 8192 - 2) == 0 and f32(1, 2 * 5000) == 0 and f32(1, 2 * 10000) == 1
        assert f64(1, 2 * 16384) == 1 and f64(1, 2 * 10000) == 0 and f64(1, 2 * 100003) == 1
        assert f32(1, 2 * 12289) == 1 and f32(1, 2 * 12289 + 1) == 0         # any length above; odd scalar counts never
        for k in range(4):
            assert put(k, 0) == 0
        assert f32(1, 1) == 0 and f32(1, 3) == 0
        assert f32(1, 2) == 1 and f32(1, 2 * 4096) == 1 and f32(1, 2 * 1000) == 1 and f64(1, 2 * 7) == 1
        assert put(L.B1_FFT_MIN_LEN_F64, 100) == 0 and get(L.B1_FFT_MIN_LEN_F64) == 100 and f64(1, 98) == 0 and f64(1, 100) == 1
        assert put(7, 1) != 0 and get(7) == 0                               # unknown key
    finally:
        for k, v in enumerate(defaults):
            put(k, v)


def test_the_product_library_reads_no_environment_variable():
    """Experiment switches live in the LAB build only (make -C basic_dsp_amd/csrc lab): the shipped library does not even
    import getenv."""
    import subprocess
    import basic_dsp_amd._lib as L
    if os.path.basename(L.LIB_PATH) != "libbasic_dsp_hip.so":
        pytest.skip("a library override is loaded")
    out = subprocess.run(["nm", "-D", "--undefined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in out


def test_fft_passes_and_block_share_argument_check_without_a_gpu():
    """Plan queries answer without a device: trips through memory of a power-of-two transform (1 resident, 2 / 3 global
    passes, 0 = not a power-of-two plan).  The per-call block shares of bdsp_hip_dev_convolve_ex are checked before
    anything touches a device, and the old process-wide knob is gone from the ABI."""
    import basic_dsp_amd as b
    f = b.lib.bdsp_hip_fft_passes
    for elem in (0, 1):
        assert f(elem, 16) == 1 and f(elem, 4096) == 1
        assert f(elem, 1 << 13) == (1 if elem == 0 else 2)  # f32: the one-workgroup 8192-point kernel (fft_pow2 -> launch_wg4)
        assert f(elem, 1 << 14) == 2 and f(elem, 1 << 20) == 2 and f(elem, 1 << 22) == 2
        assert f(elem, 1 << 23) == 3 and f(elem, 1 << 24) == 3 and f(elem, 1 << 30) == 3
        assert f(elem, 0) == 0 and f(elem, 1000) == 0 and f(elem, 1 << 31) == 0
    assert not hasattr(b.lib, "bdsp_hip_conv_block_shares")
    ex = b.lib.bdsp_hip_dev_convolve_ex
    for bad in ((60, 45), (0, 10), (-1, 5)):
        assert ex(0, None, None, 0, 0, None, 0, bad[0], bad[1], None) == 7  # BDSP_ERR_ARG_LENGTH, before any device call
        assert b"shares" in b.lib.bdsp_hip_last_error()


def test_overlap_discard_reports_failure_out_of_band_without_a_gpu():
    """The B1 return value is a position, so a failure must leave a message in last_error on EVERY failing path -- also
    when the device probe failed on an earlier call or another thread: without a device every call raises."""
    import numpy as np
    import threading
    import basic_dsp_amd as b
    import basic_dsp_amd.vector as V
    if b.lib.bdsp_hip_has_gpu_support_f32():
        pytest.skip("a GPU is present")
    x = np.zeros(2 * 64, dtype=np.float32)
    tmp = np.zeros(2 * 16, dtype=np.float32)
    h = np.zeros(2 * 16, dtype=np.float32)
    for _ in range(3):
        with pytest.raises(b.BackendError):
            V.gpu_overlap_discard(x, tmp, h, 2 * 4, 2 * 8)
    seen = []

    def other_thread():
        try:
            V.gpu_overlap_discard(x, tmp, h, 2 * 4, 2 * 8)
            seen.append("returned")
        except b.BackendError as e:
            seen.append(str(e))
    t = threading.Thread(target=other_thread)
    t.start()
    t.join()
    assert seen and seen[0] != "returned" and "no HIP device" in seen[0], seen


def test_host_sim_of_workgroup_fft(tmp_path):
    """The kernels' index math, butterflies and twiddle conventions (fft_core.h) run on the CPU."""
    import subprocess
    exe = str(tmp_path / "sim_fft")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe,
                           os.path.join(ROOT, "tests", "host_sim", "sim_fft.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout


def test_header_is_plain_c_and_a_c_client_links():
    """include/basic_dsp_hip.h must be consumable by a C compiler (the reference's foreign callers are C
    and ctypes), and a plain-C client must link against the library without a GPU present."""
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "tests", "c_abi", "facade_demo.c")
    libdir = os.path.join(ROOT, "basic_dsp_amd", "lib")
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-D_GNU_SOURCE", "-I", os.path.join(ROOT, "include"),
                               src, "-L", libdir, "-lbasic_dsp_hip", "-lm", "-o", os.path.join(d, "demo")])


def test_library_exports_nothing_but_the_declared_c_abi():
    """No C++ internals leak out of the shared object: the dynamic symbol table holds the header's functions only
    (the link uses the version script tools/gen_exports.py writes from the header)."""
    import subprocess
    import basic_dsp_amd._lib as L
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()}
    extra = sorted(exported - set(declared_functions()))
    assert not [n for n in extra if n.startswith("_Z")], extra[:5]
    assert not extra, extra[:10]


def test_rust_shim_is_what_integration_md_prints_and_its_markers_are_the_cpu_arms():
    """shim/hip.rs has never met a compiler (no rustc here), so it is kept literal where it can be: INTEGRATION.md section 1
    reproduces it verbatim, the marker types are the ones of the reference's CPU arm (two empty traits over `Float` with
    blanket impls, gpu_support/fallback.rs:8-24 -- not invented associated types), the imports follow fallback.rs:3-6, and
    every extern it declares is a symbol the library exports."""
    import re
    import basic_dsp_amd as b
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = open(os.path.join(root, "shim", "hip.rs")).read()
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    body = shim[shim.index("// (imports as"):].rstrip("\n")
    assert body in md
    for line in ("use crate::numbers::*;", "pub trait GpuFloat: Float {}", "pub trait GpuRegTrait: Float {}",
                 "impl<T> GpuFloat for T where T: Float {}", "impl<T> GpuRegTrait for T where T: Float {}",
                 "pub type Gpu32 = f32;", "pub type Gpu64 = f64;", "impl<T: RealNumber> GpuSupport<T> for T {"):
        assert line in shim, line
    assert "type Reg" not in shim and "crate::RealNumber" not in shim
    externs = re.findall(r"fn (bdsp_hip_\w+)\(", shim[shim.index('extern "C"'):shim.index("fn last_error")])
    assert len(externs) == 11
    for name in externs:
        assert hasattr(b.lib, name), name


def test_no_kernel_of_the_shipped_library_has_a_private_segment(tmp_path):
    """`No kernel of the library uses scratch` checked on the shipped binary itself: the gfx950 code objects are
    cut out of the library's .hip_fatbin section (clang offload bundles) and every kernel's metadata must say
    .private_segment_fixed_size: 0 and no dynamic stack.  (Round 5: the three-pass mixed-radix kernel first shipped with 52-68
    reserved bytes per lane -- scalar spills hipcc planned for memory and then kept in vector lanes after all; nothing in
    tests/ saw it, `make resources` did.)"""
    import re
    import shutil
    import struct
    import subprocess
    import basic_dsp_amd._lib as L
    llvm = "/opt/rocm/lib/llvm/bin"
    objcopy, readelf = os.path.join(llvm, "llvm-objcopy"), os.path.join(llvm, "llvm-readelf")
    if not (os.path.exists(objcopy) and os.path.exists(readelf)):
        pytest.skip("llvm-objcopy / llvm-readelf not found")
    if os.path.basename(L.LIB_PATH) != "libbasic_dsp_hip.so":
        pytest.skip("a library override is loaded")
    fat = tmp_path / "fat.bin"
    subprocess.run([objcopy, "--dump-section", ".hip_fatbin=%s" % fat, L.LIB_PATH, str(tmp_path / "copy.so")], check=True)
    blob = fat.read_bytes()
    kernels, offenders = 0, []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        p = m.start()
        count = struct.unpack_from("<Q", blob, p + 24)[0]
        off = p + 32
        for _ in range(count):
            o, size, tl = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tl].decode()
            off += tl
            if "gfx950" not in triple or size == 0:
                continue
            co = tmp_path / ("co%d.elf" % kernels)
            co.write_bytes(blob[p + o:p + o + size])
            notes = subprocess.run([readelf, "--notes", str(co)], capture_output=True, text=True, check=True).stdout
            names = re.findall(r"\.name:\s+(\S+)", notes)
            sizes = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)
            stacks = re.findall(r"\.uses_dynamic_stack:\s+(\w+)", notes)
            kernels += len(sizes)
            kn = [n for n in names if n.startswith("_Z")]
            for i, sz in enumerate(sizes):
                if int(sz) != 0:
                    offenders.append((kn[i] if i < len(kn) else "?", int(sz)))
            assert "true" not in stacks
    assert kernels >= 600, kernels
    assert not offenders, offenders[:10]
