// shim/hip.rs -- the Rust side of the B1 boundary: `impl GpuSupport<T> for T` over libbasic_dsp_hip.so.
//
// Drop this file into the reference as vector/src/gpu_support/hip.rs and select it in
// vector/src/gpu_support/mod.rs:1-11 next to the OpenCL (`mod ocl`) and CPU (`mod fallback`) arms:
//
//     #[cfg(feature = "use_hip")] mod hip;
//     #[cfg(feature = "use_hip")] pub use self::hip::*;
//
// It replaces vector/src/gpu_support/ocl/mod.rs:133-522 (the OpenCL / clFFT backend) function for function; the C
// prototypes it binds are declared in include/basic_dsp_hip.h (B1 section), each citing the trait method it stands for
// (vector/src/gpu_support/mod.rs:18-46).  Link with `-lbasic_dsp_hip` (build: make -C basic_dsp_amd/csrc).
// NOT compiled in this repository's image (no rustc / cargo here): the ABI it binds is exercised from C
// (tests/c_abi/facade_demo.c) and Python (tests/test_abi.py, tests/test_gpu_parity.py: test_b1_*).
// (imports as vector/src/gpu_support/fallback.rs:3-6 -- `RealNumber` and `Float` both come out of crate::numbers --
// plus what the FFI needs)
use super::GpuSupport;
use crate::numbers::*;
use std::ops::Range;
use rustfft::FftDirection;
use std::mem;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

#[link(name = "basic_dsp_hip")]
extern "C" {
    fn bdsp_hip_has_gpu_support_f32() -> c_int;
    fn bdsp_hip_has_gpu_support_f64() -> c_int;
    fn bdsp_hip_is_supported_fft_len_f32(is_complex: c_int, len: usize) -> c_int;
    fn bdsp_hip_is_supported_fft_len_f64(is_complex: c_int, len: usize) -> c_int;
    fn bdsp_hip_fft_f32(is_complex: c_int, signal: *mut f32, len: usize, inverse: c_int) -> c_int;
    fn bdsp_hip_fft_f64(is_complex: c_int, signal: *mut f64, len: usize, inverse: c_int) -> c_int;
    fn bdsp_hip_convolve_vector_f32(is_complex: c_int, src: *const f32, src_len: usize, dst: *mut f32,
        dst_len: usize, imp: *const f32, imp_len: usize, range_start: *mut usize, range_end: *mut usize) -> c_int;
    fn bdsp_hip_convolve_vector_f64(is_complex: c_int, src: *const f64, src_len: usize, dst: *mut f64,
        dst_len: usize, imp: *const f64, imp_len: usize, range_start: *mut usize, range_end: *mut usize) -> c_int;
    fn bdsp_hip_overlap_discard_f32(x_time: *mut f32, x_len: usize, tmp: *mut f32, tmp_len: usize,
        x_freq: *mut f32, x_freq_len: usize, h_freq: *const f32, h_len: usize, imp_len: usize, step_size: usize) -> usize;
    fn bdsp_hip_overlap_discard_f64(x_time: *mut f64, x_len: usize, tmp: *mut f64, tmp_len: usize,
        x_freq: *mut f64, x_freq_len: usize, h_freq: *const f64, h_len: usize, imp_len: usize, step_size: usize) -> usize;
    fn bdsp_hip_last_error() -> *const c_char;   // per-thread; overlap_discard clears it on entry
}

/// The backend's error channel: a position is a position (0 included), a failure is a non-empty message.
fn last_error() -> String {
    unsafe { CStr::from_ptr(bdsp_hip_last_error()).to_string_lossy().into_owned() }
}

// The marker types the rest of the crate names (vector/src/lib.rs:110,162,183,204,208-209: `type GpuReg = Gpu32 / Gpu64`,
// `RealNumber: Float + DspNumber + GpuFloat + ...`).  This backend needs nothing from them -- it takes the caller's
// slices as they are -- so they are EXACTLY the CPU arm's (vector/src/gpu_support/fallback.rs:8-24): plain scalars and two
// empty traits over `Float` with blanket impls.  (The OpenCL arm binds them to ocl / clFFT types instead,
// ocl/mod.rs:20-37.)
pub type Gpu32 = f32;

pub type Gpu64 = f64;

pub trait GpuFloat: Float {}

pub trait GpuRegTrait: Float {}

impl<T> GpuFloat for T where T: Float {}

impl<T> GpuRegTrait for T where T: Float {}

impl<T: RealNumber> GpuSupport<T> for T {
    fn has_gpu_support() -> bool {                       // gpu_support/mod.rs:21
        unsafe { if mem::size_of::<T>() == 4 { bdsp_hip_has_gpu_support_f32() != 0 }
                 else { bdsp_hip_has_gpu_support_f64() != 0 } }
    }
    fn is_supported_fft_len(is_complex: bool, len: usize) -> bool {   // mod.rs:32
        unsafe { if mem::size_of::<T>() == 4 { bdsp_hip_is_supported_fft_len_f32(is_complex as c_int, len) != 0 }
                 else { bdsp_hip_is_supported_fft_len_f64(is_complex as c_int, len) != 0 } }
    }
    fn fft(is_complex: bool, signal: &mut [T], direction: FftDirection) {   // mod.rs:35
        let inv = (direction == FftDirection::Inverse) as c_int;
        let rc = unsafe { if mem::size_of::<T>() == 4 {
            bdsp_hip_fft_f32(is_complex as c_int, signal.as_mut_ptr() as *mut f32, signal.len(), inv)
        } else {
            bdsp_hip_fft_f64(is_complex as c_int, signal.as_mut_ptr() as *mut f64, signal.len(), inv)
        } };
        assert!(rc == 0, "HIP fft failed with code {}", rc);        // the OpenCL impl panics too
    }
    fn gpu_convolve_vector(is_complex: bool, source: &[T], target: &mut [T], imp_resp: &[T])
        -> Option<Range<usize>> {                                    // mod.rs:24-29
        let (mut s, mut e) = (0usize, 0usize);
        let rc = unsafe { if mem::size_of::<T>() == 4 {
            bdsp_hip_convolve_vector_f32(is_complex as c_int, source.as_ptr() as *const f32, source.len(),
                target.as_mut_ptr() as *mut f32, target.len(), imp_resp.as_ptr() as *const f32, imp_resp.len(), &mut s, &mut e)
        } else {
            bdsp_hip_convolve_vector_f64(is_complex as c_int, source.as_ptr() as *const f64, source.len(),
                target.as_mut_ptr() as *mut f64, target.len(), imp_resp.as_ptr() as *const f64, imp_resp.len(), &mut s, &mut e)
        } };
        assert!(rc >= 0, "HIP convolution failed with code {}", rc);
        if rc == 1 { Some(s..e) } else { None }
    }
    fn overlap_discard(x_time: &mut [T], tmp: &mut [T], x_freq: &mut [T], h_freq: &[T],
                       imp_len: usize, step_size: usize) -> usize {  // mod.rs:38-45
        let pos = unsafe { if mem::size_of::<T>() == 4 {
            bdsp_hip_overlap_discard_f32(x_time.as_mut_ptr() as *mut f32, x_time.len(), tmp.as_mut_ptr() as *mut f32, tmp.len(),
                x_freq.as_mut_ptr() as *mut f32, x_freq.len(), h_freq.as_ptr() as *const f32, h_freq.len(), imp_len, step_size)
        } else {
            bdsp_hip_overlap_discard_f64(x_time.as_mut_ptr() as *mut f64, x_time.len(), tmp.as_mut_ptr() as *mut f64, tmp.len(),
                x_freq.as_mut_ptr() as *mut f64, x_freq.len(), h_freq.as_ptr() as *const f64, h_freq.len(), imp_len, step_size)
        } };
        // Failure is reported OUT OF BAND: the return value is a position, and every position -- 0 included -- is a
        // legal one (convolution.rs:400-412 continues the scalar tail from wherever the blocks stopped).
        let err = last_error();
        assert!(err.is_empty(), "HIP overlap_discard failed: {}", err);
        pos
    }
}
