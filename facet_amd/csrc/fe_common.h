// facet_amd engine — shared declarations for the HIP kernels and the C-ABI layer.
// gfx950 (MI355X) only. No torch types anywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include "fe_check.h"
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace fe {


#define FE_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      char _b[512];                                                                       \
      snprintf(_b, sizeof _b, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,               \
               hipGetErrorString(_e));                                                    \
      throw fe::Error(_b);                                                                \
    }                                                                                     \
  } while (0)

// hipFuncSetAttribute applies to the CURRENT device only; `done` (one per kernel instantiation) keeps a bit per device ordinal,
// so a process that drives several GPUs (one context each) raises the dynamic-LDS limit on each of them once.
inline void ensure_dynamic_lds(const void* kernel, size_t bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  FE_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  FE_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  done.fetch_or(bit, std::memory_order_release);
}

// Activation codes shared by the GEMM/conv epilogue.
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SIGMOID = 3, ACT_PRELU = 4, ACT_SOFTPLUS = 5 };  // PRELU: per-channel slope

// Scalar activations of the contraction epilogues (PReLU is handled by the callers: it needs the per-channel slope).
// Softplus follows torch.nn.Softplus(beta=1, threshold=20): x above the threshold passes through.
__device__ __forceinline__ float fe_apply_act(float v, int act) {
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  if (act == ACT_SOFTPLUS) return v > 20.f ? v : log1pf(expf(v));
  return v;
}

__device__ __forceinline__ float fe_rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32, 1 ulp (see fe_gelu_fast)

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute): ~12 instructions against ocml's erff (~40 with its branches). The GELU
// built on it is within 1e-7 * |x| of the erf form - used where a result is NOT rounded to 2 bytes afterwards but fp32's last bits do
// not matter either (the split-pair and fp32-stream epilogues of the 2-byte kernels; the fp32 kernels keep erff).
__device__ __forceinline__ float fe_erf_as(float x) {
  const float ax = fabsf(x);
  const float t = fe_rcp_fast(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float fe_apply_act_precise(float v, int act) {      // fe_apply_act with the A&S erf
  if (act == ACT_GELU) return 0.5f * v * (1.f + fe_erf_as(v * 0.70710678118654752440f));
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  if (act == ACT_SOFTPLUS) return v > 20.f ? v : log1pf(expf(v));
  return v;
}

// GELU for epilogues whose result is rounded to bf16: the tanh form 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3), written as
// x * sigmoid(2u) = x / (1 + 2^(-2u log2 e)): 3 multiply-adds, one v_exp, one add, one v_rcp, one multiply. It deviates from the erf
// form by at most 4.8e-4 (at |x| ~ 2.7, where a bf16 ulp is 1.6e-2) and by < 2e-5 for |x| < 0.5 - below the rounding of the store.
// The exponential and the reciprocal are the bare v_exp_f32 / v_rcp_f32 (1 ulp each): exp2f() wraps the instruction in a denormal
// rescue (compare, two selects, add, ldexp) and __frcp_rn() expands to the IEEE division sequence (div_scale, rcp, four FMAs, div_fmas,
// div_fixup) - ~30 instructions per value where 8 do; a flushed denormal 2^z only turns 1 + 2^z into 1, and z = +inf gives x * 0.
__device__ __forceinline__ float fe_gelu_fast(float x) {
  const float x2 = x * x;
  const float z = x * (-2.3022082f - 0.10294324f * x2);       // -2 u log2(e)
  return x * fe_rcp_fast(1.0f + __builtin_amdgcn_exp2f(z));
}
__device__ __forceinline__ float fe_apply_act_fast(float v, int act) {   // bf16 epilogues only
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_GELU) return fe_gelu_fast(v);
  if (act == ACT_SIGMOID) return fe_rcp_fast(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
  if (act == ACT_SOFTPLUS) return fmaxf(v, 0.f) + __logf(1.f + __expf(-fabsf(v)));   // |error| ~1e-7: far below the bf16 rounding of the result
  return v;
}

// bf16 storage type of the reduced-precision path (BASELINE configs[3]). hipcc lowers float <-> __bf16 casts to
// v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN) / a 16-bit shift.
typedef __bf16 bf16;
// fp16 storage type: the reference's own reduced precision (`self.model.half()`, processing/scorer.py:513-516). 11 significant bits
// against bf16's 8, the same matrix-core rate on gfx950 (v_mfma_f32_32x32x16_f16); the range ends at 65504, so every fp32 -> fp16
// store of the engine saturates instead of producing an infinity.
typedef _Float16 f16;
constexpr float FE_F16_MAX = 65504.f;
__device__ __forceinline__ f16 fe_to_f16(float v) { return (f16)__builtin_fminf(__builtin_fmaxf(v, -FE_F16_MAX), FE_F16_MAX); }

// The two 2-byte element types share every kernel of the reduced-precision path through these overloads (tag = a null pointer of
// the element type): unpack a 32-bit pair to two floats, pack two floats to a pair (round to nearest even).
typedef __bf16 fe_bf2 __attribute__((ext_vector_type(2)));
typedef _Float16 fe_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fe_unpack2(const bf16*, unsigned u, float& lo, float& hi) {
  lo = __uint_as_float(u << 16); hi = __uint_as_float(u & 0xFFFF0000u);
}
__device__ __forceinline__ void fe_unpack2(const f16*, unsigned u, float& lo, float& hi) {
  const fe_h2 v = __builtin_bit_cast(fe_h2, u);
  lo = (float)v[0]; hi = (float)v[1];
}
__device__ __forceinline__ unsigned fe_pack2(const bf16*, float lo, float hi) {
  fe_bf2 v; v[0] = (bf16)lo; v[1] = (bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned fe_pack2(const f16*, float lo, float hi) {
  fe_h2 v; v[0] = fe_to_f16(lo); v[1] = fe_to_f16(hi);
  return __builtin_bit_cast(unsigned, v);
}
// one 32x32x16 matrix instruction on 8-element fragments held as four dwords
typedef float fe_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 fe_bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 fe_h8 __attribute__((ext_vector_type(8)));
typedef float fe_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ fe_f32x16 fe_mfma16(const bf16*, fe_v4f a, fe_v4f b, fe_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fe_bf8, a), __builtin_bit_cast(fe_bf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ fe_f32x16 fe_mfma16(const f16*, fe_v4f a, fe_v4f b, fe_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fe_h8, a), __builtin_bit_cast(fe_h8, b), c, 0, 0, 0);
}

// A view of an NHWC activation in HBM (T = float, or bf16 on the reduced-precision path). `ld` is the channel stride of one
// pixel (>= c) so a view can be a channel slice of a wider (concat) buffer.
template <class T>
struct TensorT {
  T* p = nullptr;
  int n = 0, h = 0, w = 0, c = 0;
  int ld = 0;
  size_t pixels() const { return (size_t)n * h * w; }
  size_t numel() const { return pixels() * c; }
  TensorT slice(int c0, int cn) const {
    TensorT t = *this;
    t.p = p + c0;
    t.c = cn;
    return t;
  }
  // the same shape with a NULL pointer of another element type: the "no 2-byte output" view of a layer that only writes fp32 (ConvOptsT::y32)
  template <class U> TensorT<U> retype() const {
    TensorT<U> t;
    t.p = nullptr; t.n = n; t.h = h; t.w = w; t.c = c; t.ld = c;
    return t;
  }
};
using Tensor = TensorT<float>;
using TensorH = TensorT<bf16>;
using TensorF16 = TensorT<f16>;

// element access shared by the fp32 and bf16 instantiations: arithmetic is always fp32, only loads / stores differ
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return (float)*p; }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16* p, float v) { *p = (bf16)v; }
__device__ __forceinline__ float ldf(const f16* p) { return (float)*p; }
__device__ __forceinline__ void stf(f16* p, float v) { *p = fe_to_f16(v); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ float4 ld4(const f16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  float4 v;
  fe_unpack2(p, u.x, v.x, v.y); fe_unpack2(p, u.y, v.z, v.w);
  return v;
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(f16* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(fe_pack2(p, v.x, v.y), fe_pack2(p, v.z, v.w));
}
__device__ __forceinline__ void st4(bf16* p, float4 v) {
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  bf4 o; o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
  *reinterpret_cast<bf4*>(p) = o;
}

// Bump allocator over one hipMalloc'ed slab; reset per forward. 256-B aligned.
class Arena {
 public:
  void init(size_t bytes) {
    release();
    FE_HIP(hipMalloc(&base_, bytes));
    cap_ = bytes;
    off_ = 0;
  }
  void release() {
    if (base_) (void)hipFree(base_);
    base_ = nullptr;
    cap_ = off_ = 0;
  }
  ~Arena() { release(); }
  void reset() { off_ = 0; }
  size_t mark() const { return off_; }
  void rewind(size_t m) { off_ = m; }
  size_t capacity() const { return cap_; }
  size_t high_water() const { return high_; }
  void* alloc(size_t bytes) {
    size_t a = (off_ + 255) & ~(size_t)255;
    FE_CHECK(a + bytes <= cap_, "arena exhausted: need %zu at %zu of %zu", bytes, a, cap_);
    off_ = a + bytes;
    if (off_ > high_) high_ = off_;
    return (char*)base_ + a;
  }
  template <class T>
  TensorT<T> tensor_t(int n, int h, int w, int c) {
    TensorT<T> t;
    t.n = n; t.h = h; t.w = w; t.c = c; t.ld = c;
    t.p = (T*)alloc(t.numel() * sizeof(T));
    return t;
  }
  Tensor tensor(int n, int h, int w, int c) { return tensor_t<float>(n, h, w, c); }
  template <class T>
  T* array(size_t count) { return (T*)alloc(count * sizeof(T)); }
 private:
  void* base_ = nullptr;
  size_t cap_ = 0, off_ = 0, high_ = 0;
};

// ---------------------------------------------------------------------------------------
// Implicit-GEMM convolution / GEMM (kernels_conv.hip)
// ---------------------------------------------------------------------------------------
template <class T>
struct ConvParamsT {
  const T* x; int ldx;              // input NHWC view
  const T* w;                       // packed [Cout][Kp], zero padded to Kp; K order: see pack_conv (fp32) / pack_conv_bf16
  const float* scale;               // per-Cout multiply (nullable => 1)
  const float* shift;               // per-Cout add (nullable => 0)
  const float* slope;               // per-Cout negative slope, ACT_PRELU only
  const T* res; int ldr;            // residual view, same N,Ho,Wo,Cout (nullable)
  const T* gate; int ldg;           // multiplicative gate, ldg==... (nullable); gate_c1: single channel broadcast
  T* y; int ldy;                    // output NHWC view
  int N, H, W, Cin, Ho, Wo, Cout;
  int KH, KW, sh, sw, ph, pw, dh, dw;
  int K, Kp, M;
  int act;                          // Act
  int res_after_act;                // 1: y = act(conv*scale+shift) + res ; 0: y = act(conv*scale+shift+res)
  int gate_c1;                      // 1: gate has one channel (broadcast over Cout)
  int vec_epi;                      // set by launch_conv: 16-B vector epilogue is legal
  int unit_stride;                  // set by the launchers: 1x1 kernel, stride 1, no padding (input pixel index == output pixel index)
  int ldw;                          // weight row stride in elements (0 => Kp); lets an activation matrix act as B
  // batched launches (gridDim.y = batch): b = bo*nb1 + bi; pointer += bo*s2 + bi*s1 (element strides)
  int batch, nb1;
  long long xs1, xs2, ws1, ws2, ys1, ys2, hs1;   // hs1: shift stride per inner index
  int buf_ok; unsigned x_span, w_span;  // set by launch_conv_dma: operands addressable through 32-bit buffer offsets
  unsigned y_span, r_span, g_span;      // set by launch_conv_bf16: byte spans of y / res / gate (0: beyond 32-bit buffer addressing)
  int variant;                      // 0 = auto tile choice; >0 forces a tile variant (tools/conv_bench.py)
  int cb;                           // bf16 kernel: channel block of the packed K order (32, or 16 for Cin % 32 != 0 spatial kernels)
  int pad_store;                    // bf16 kernel: columns [Cout, roundup8(Cout)) exist in y and are written as zeros (V^T GEMM)
  // 2-byte kernels, fp32 streams (FE_PRECISION_RES32): the residual read as fp32 (instead of `res`) and / or the result also written
  // as fp32 rows; `y` may then be null (fp32 output only). Same N, Ho, Wo, Cout as y.
  const float* res32; int ldr32;
  float* y32; int ldy32;
  unsigned r32_span, y32_span;      // set by launch_conv_bf16: byte spans for buffer addressing
  int a_wrap;                       // 2-byte GEMM form: > 0 = the A row holds only a_wrap * 32 columns and is read again from column 0 after them
  int split_lo_off;                 // 2-byte fp32-stream forms: > 0 = split-pair output, hi at column c and lo = round(v - hi) at column split_lo_off + c of y
  int exact_act;                    // 2-byte fp32-stream forms: erf GELU instead of the tanh form (results that are NOT rounded to 2 bytes)
};
using ConvParams = ConvParamsT<float>;
using ConvParamsH = ConvParamsT<bf16>;

// Packed conv weight living in HBM.
struct ConvW {
  float* w = nullptr;      // [Cout][Kp]
  float* scale = nullptr;  // [Cout] or null
  float* shift = nullptr;  // [Cout] or null
  float* slope = nullptr;  // [Cout] PReLU slopes or null
  int Cout = 0, Cin = 0, CinPad = 0, KH = 1, KW = 1, K = 0, Kp = 0;
  int CoutAlg = 0;         // logical output channels when Cout carries zero padding (FLOP accounting only; 0 => Cout)
  // Cout <= 2 spatial kernels also carry a tap-decomposed form: rows (tap, co) of a 1x1 conv [KH*KW*Cout][KpT]
  float* wtap = nullptr;
  int wino_m = 0;          // 2 or 4: which Winograd form `wino` holds (F(2x2,3x3): 16 planes, F(4x4,3x3): 36 planes)
  float* wino = nullptr;   // 3x3 convs with Cin >= 256 also carry Winograd F(2x2,3x3) weights U[16][Cout][Cin]
  float* wstem = nullptr;  // 7x7 or 3x3, Cin <= 3, Cout 32|64 stems also carry the [taps][Cout][4] layout of kernels_stem.hip
  float* wn16 = nullptr;   // fp32 3x3 layers with 16 output channels and 16 / 32 / 64 input channels also carry the layout of kernels_n16.hip
  int KpT = 0;
  // 2-byte path (models committed under FE_PRECISION_BF16 / FE_PRECISION_F16): [Cout][KpH] elements of type `hprec` in the K
  // order of kernels_conv_bf16.hip
  void* wh = nullptr;
  int hprec = 0;           // element type of wh / wtap_h: 1 bf16, 2 f16 (Precision codes); the launch wrappers check it against T
  int KpH = 0, cb = 0, CinPadH = 0;   // CinPadH: input channels the bf16 kernel reads per pixel (Cin rounded up to 8)
  void* wtap_h = nullptr;  // tap-decomposed form for Cout <= 2 spatial kernels, [KH*KW*Cout rounded to 8][KpTH]
  int KpTH = 0;
};

template <class T>
struct ConvOptsT {
  int sh = 1, sw = 1, ph = 0, pw = 0, dh = 1, dw = 1;
  int act = ACT_NONE;
  const TensorT<T>* res = nullptr;
  int res_after_act = 0;
  const TensorT<T>* gate = nullptr;
  // 2-byte layers of a model with fp32 residual streams: residual taken from / result also delivered to an fp32 tensor (the TensorT<T>
  // output view passed to conv_forward may then carry a null pointer: fp32 output only)
  const TensorT<float>* res32 = nullptr;
  const TensorT<float>* y32 = nullptr;
};
using ConvOpts = ConvOptsT<float>;

constexpr int CONV_KALIGN = 32;  // Kp is a multiple of this (covers BK = 16 and 32)
constexpr int CONV_KALIGN_H = 64; // bf16 kernel: K padded to whole K-steps of two 32-element slabs

void launch_conv(const ConvParams& p, hipStream_t s);
// Winograd transforms (kernels_winograd.hip): V / M are [16][tiles][C] planes
void launch_wino_input(const Tensor& x, int th, int tw, float* V, hipStream_t s);
void launch_wino_output(const float* Mb, const Tensor& y, int th, int tw, const float* scale, const float* shift, int relu, hipStream_t s);
void launch_wino4_input(const Tensor& x, int th, int tw, float* V, hipStream_t s);     // F(4x4,3x3): [36][tiles][C] planes
void launch_wino4_output(const float* Mb, const Tensor& y, int th, int tw, const float* scale, const float* shift, int act, const float* slope,
                         const Tensor* res, int res_after_act, hipStream_t s);
template <class TO>   // TO = float | bf16: the stem always reads fp32 NHWC4 pixels; it can hand bf16 activations to the bf16 path
bool launch_stem(const float* x, int ldx, int N, int H, int W, const float* wstem, const float* scale, const float* shift, const float* slope,
                 int Cout, int k, int stride, int act, TO* y, int ldy, int Ho, int Wo, hipStream_t s);   // kernels_stem.hip; false = shape not handled
void launch_conv_narrow(const ConvParams& p, hipStream_t s);            // Cout <= 4, no MFMA (kernels_misc.hip)
void launch_conv_dma(const ConvParams& p, int tile, hipStream_t s);  // LDS-DMA fast path (kernels_conv_dma.hip)
void launch_conv_bf16(const ConvParamsH& p, hipStream_t s);          // bf16 MFMA implicit GEMM (kernels_conv_bf16.hip)
void launch_conv_bf16(const ConvParamsT<f16>& p, hipStream_t s);     // the same kernels on fp16 elements (kernels_conv_f16.hip)
// A convolution over a batch of images whose input, output or residual tensor outgrows the 32-bit buffer addressing of the LDS-DMA
// kernels (4 GiB: e.g. 64 fp32 images of 512 x 512 x 64) is issued as several launches over image groups that fit. Returns false when the
// problem is not an image batch (GEMMs, batched launches) or a single image is already too large.
template <class T, class Launch>
inline bool conv_split_by_images(const ConvParamsT<T>& p, Launch&& launch) {
  if (p.batch > 1 || p.N <= 1 || (long long)p.N * p.Ho * p.Wo != (long long)p.M) return false;
  const unsigned long long in_img = (unsigned long long)p.H * p.W * p.ldx * sizeof(T);
  unsigned long long out_img = (unsigned long long)p.Ho * p.Wo * (unsigned long long)(p.ldy > p.ldr ? p.ldy : p.ldr) * sizeof(T);
  const unsigned long long out32_img = (unsigned long long)p.Ho * p.Wo * (unsigned long long)(p.ldy32 > p.ldr32 ? p.ldy32 : p.ldr32) * sizeof(float);
  if ((p.y32 || p.res32) && out32_img > out_img) out_img = out32_img;
  const unsigned long long per_img = in_img > out_img ? in_img : out_img;
  constexpr unsigned long long LIMIT = 0xF0000000ull;
  if (per_img >= LIMIT) return false;
  const int ng = (int)(LIMIT / per_img);
  if (ng >= p.N) return false;
  for (int i0 = 0; i0 < p.N; i0 += ng) {
    ConvParamsT<T> q = p;
    const int n = ng < p.N - i0 ? ng : p.N - i0;
    const size_t opix = (size_t)i0 * p.Ho * p.Wo;
    q.N = n; q.M = n * p.Ho * p.Wo;
    q.x = p.x + (size_t)i0 * p.H * p.W * p.ldx;
    if (p.y) q.y = p.y + opix * p.ldy;
    if (p.y32) q.y32 = p.y32 + opix * p.ldy32;
    if (p.res32) q.res32 = p.res32 + opix * p.ldr32;
    if (p.res) q.res = p.res + opix * p.ldr;
    if (p.gate) q.gate = p.gate + opix * p.ldg;
    launch(q);
  }
  return true;
}

double conv_flops(const ConvParams& p);

// ---------------------------------------------------------------------------------------
// Misc kernels (kernels_misc.hip)
// ---------------------------------------------------------------------------------------
// u8 HWC RGB -> fp32 NHWC4 ((v/255 - mean)/std, 4th channel 0). bgr: input channel order is BGR.
void launch_u8_to_nhwc4_norm(const uint8_t* src, float* dst, size_t pixels, const float mean[3],
                             const float stdv[3], int bgr, hipStream_t s);
// The helper kernels below are templates over the activation element type T (float | bf16; arithmetic is fp32 in both),
// explicitly instantiated in kernels_misc.hip.
// fp32 NCHW -> NHWC(c padded to cpad with zeros)
template <class T> void launch_nchw_to_nhwc(const float* src, T* dst, int n, int c, int h, int w, int cpad, hipStream_t s);
template <class T> void launch_nhwc_to_nchw(const T* src, int lds_, float* dst, int n, int c, int h, int w, hipStream_t s);
// max pool, NHWC; ceil_mode handled by caller through Ho/Wo; padding implicit -inf.
template <class T> void launch_maxpool(const TensorT<T>& x, const TensorT<T>& y, int k, int stride, int pad, hipStream_t s);
// bilinear resize (align_corners=False), NHWC.
template <class T> void launch_bilinear(const TensorT<T>& x, const TensorT<T>& y, hipStream_t s);
// adaptive average pool NHWC -> NHWC (torch semantics: start=floor(i*H/Ho), end=ceil((i+1)*H/Ho))
template <class T> void launch_adaptive_avgpool(const TensorT<T>& x, const TensorT<T>& y, hipStream_t s);
void launch_convert(const float* x, bf16* y, size_t n, hipStream_t s);
void launch_convert(const bf16* x, float* y, size_t n, hipStream_t s);
void launch_convert(const float* x, f16* y, size_t n, hipStream_t s);
void launch_convert(const f16* x, float* y, size_t n, hipStream_t s);
void launch_splitk_reduce(const float* part, int splits, int M, int N, const float* scale, const float* shift, const float* slope, int act,
                          const float* res, int ldr, int res_after_act, float* y, int ldy, hipStream_t s);
// y = act(x) elementwise / y = x + r
// y[n,oh,ow,co] = act(scale*sum_tap z[n, oh-ph+kh*dh, ow-pw+kw*dw, tap*Cout+co] + shift)  (stride 1)
template <class T>
void launch_tap_gather(const T* z, int ldz, int n, int h, int w, int kh, int kw, int ph, int pw, int dh, int dw,
                       int cout, const float* scale, const float* shift, int act, T* y, int ldy, int ho, int wo,
                       hipStream_t s);
// fused attention, head_dim 64 (kernels_attn.hip): o = softmax(q k^T) v + bv per (batch, head)
void launch_attention(const float* q, int ldq, const float* k, int ldk, const float* vt, int lp, const float* bv, float* o,
                      int ldo, int B, int H, int Lq, int Lk, int dmodel, int causal, hipStream_t s);
// 2-byte twins (kernels_attn_bf16.hip): bf16 or fp16 Q / K / V^T / O, fp32 softmax and accumulation
void launch_attention(const bf16* q, int ldq, const bf16* k, int ldk, const bf16* vt, int lp, const float* bv, bf16* o,
                      int ldo, int B, int H, int Lq, int Lk, int dmodel, int causal, hipStream_t s);
void launch_attention(const f16* q, int ldq, const f16* k, int ldk, const f16* vt, int lp, const float* bv, f16* o,
                      int ldo, int B, int H, int Lq, int Lk, int dmodel, int causal, hipStream_t s);
// split-operand attention (kernels_attn_split.hip): q / k rows [.][ld] with hi at column c and lo at lo_off + c, V^T as two planes,
// output rows [.][ldo] as hi | lo (o_lo_off)
void launch_attention_split(const f16* q, const f16* k, int ld, int lo_off, const f16* vt_hi, const f16* vt_lo, int lp, f16* o, int ldo, int o_lo_off,
                            int B, int H, int Lq, int Lk, int dmodel, hipStream_t s);
// M <= 32 rows: one wave per output column (kernels_misc.hip); TI / TW / TO = activation / weight / output element types
template <class TI, class TW, class TO>
void launch_gemm_skinny(const TI* x, int ldx, const TW* w, int ldw, const float* scale, const float* shift, TO* y,
                        int ldy, int M, int N, int K, int act, hipStream_t s);
void launch_sigmoid(const Tensor& x, const Tensor& y, hipStream_t s);
void launch_add(const Tensor& a, const Tensor& b, const Tensor& y, hipStream_t s);
// LayerNorm over last dim of a [rows][d] matrix (eps inside sqrt, biased variance); statistics in fp32
// LayerNorm of fp32 rows written as fp16 pairs: y[row] = [hi(d) | lo(d)], hi = fp16(v), lo = fp16(v - hi) (ldy >= 2 d); d = 1024 or 768
void launch_layernorm_split(const float* x, int ldx, f16* y, int ldy, const float* g, const float* b, int rows, int d, float eps, hipStream_t s);
// x fp32 [rows][cols] -> y fp16 [rows][2 cols] = [hi | lo]
void launch_split_hi_lo(const float* x, f16* y, size_t rows, int cols, hipStream_t s);
template <class T, class TO>   // TO = T, or a 2-byte TO behind fp32 rows (the fp32 residual stream of the reduced-precision path)
void launch_layernorm(const T* x, int ldx, TO* y, int ldy, const float* g, const float* b,
                      int rows, int d, float eps, hipStream_t s);
// rows softmax in place for [rows][d]
void launch_softmax_rows(float* x, int ld, int rows, int d, hipStream_t s);
// same, and zero-fills columns [d, ld) so the padded tail contributes nothing to P.V
void launch_softmax_rows_pad(float* x, int ld, int rows, int d, hipStream_t s);
// y[rows][d] += pos[(row % L)][d]
template <class T>
void launch_add_rows_bcast(T* y, int ldy, const float* pos, int rows, int L, int d, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Graph-runtime kernels (kernels_graph.hip); channel counts are multiples of 4
// ---------------------------------------------------------------------------------------
void launch_affine_act(const Tensor& x, const Tensor& y, const float* scale, const float* shift, int act, const float* slope,
                       hipStream_t s);
// op: 0 add, 1 sub, 2 mul, 3 div
void launch_binary(const Tensor& a, const Tensor& b, const Tensor& y, int op, int act, hipStream_t s);
// dst dense [dims0..5] row-major; element (i0..i5) read from src[sum i_k * sstr_k]
void launch_gather_strided(const float* src, float* dst, const long long dims[6], const long long sstr[6], hipStream_t s);
void launch_nearest(const Tensor& x, const Tensor& y, hipStream_t s);
void launch_avgpool(const Tensor& x, const Tensor& y, int k, int stride, int pad, int include_pad, hipStream_t s);
// wt: [kh*kw][C] tap-major depthwise weights
void launch_dwconv(const Tensor& x, const Tensor& y, const float* wt, int kh, int kw, int sh, int sw, int ph, int pw,
                   const float* scale, const float* shift, int act, const float* slope, const Tensor* res, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Face-path pixel kernels (kernels_face.hip)
// ---------------------------------------------------------------------------------------
void launch_cv_resize_linear(const uint8_t* src, int n, int h, int w, uint8_t* dst, int ch, int cw, int nh, int nw, const int* xofs,
                             const short* ialpha, const int* yofs, const short* ibeta, int area2, hipStream_t s);
void launch_warp_affine(const uint8_t* src, int h, int w, const int* img_of, const double* Minv, int m, int S, const short* wtab,
                        uint8_t* dst, hipStream_t s);
void launch_swap_rb_u8(const uint8_t* src, uint8_t* dst, size_t pixels, hipStream_t s);
void launch_u8_blob(const uint8_t* src, float* dst, size_t pixels, float mean, float scale, int swap_rb, hipStream_t s);
void launch_scrfd_decode(const float* scores, const float* bbox, const float* kps, int n, int fh, int fw, int A, int K, int stride,
                         float thresh, float det_scale, int level, float* cand, int* counts, int max_cand, hipStream_t s);
void cv_resize_tables(int src, int dst, bool clamp_fx, std::vector<int>& ofs, std::vector<short>& coef);
void cv_warp_weight_table(std::vector<short>& wtab);

// ---------------------------------------------------------------------------------------
// Per-image technical statistics (kernels_stats.hip)
// ---------------------------------------------------------------------------------------
constexpr int FE_STATS_COUNT = 264;   // doubles per image: [0..255] gray histogram, 256 lap sum, 257 lap sum sq, 258 sum|Immerkaer|, 259 sum S, 260 sum c*log2(c) over the H-S histogram
size_t stats_accum_bytes(int n);
void launch_image_stats(const uint8_t* bgr, int n, int h, int w, uint8_t* gray, uint8_t* hsv_out, const int* sdiv, const int* hdiv, void* accum,
                        double* out, hipStream_t s);
void cv_hsv_tables(std::vector<int>& sdiv, std::vector<int>& hdiv);
// rois: device int [m][4] = x1,y1,x2,y2 (already clipped to the image, x2/y2 exclusive); out: device double [m][4]
void launch_roi_laplacian(const uint8_t* bgr, int h, int w, const int* img_of, const int* rois, int m, double* out, hipStream_t s);
// kernels_lines.hip: Gaussian 5x5 + Canny non-maximum-suppression map on the GPU (host stage: lines_host.h)
void launch_canny_map(const uint8_t* d_bgr, int n, int h, int w, int low, int high, uint8_t* d_blur, void* d_grad, void* d_mag, uint8_t* d_map,
                      hipStream_t s);

}  // namespace fe
