// HBM-bound row kernels of the Semi-PD hot path for gfx950:
// RMSNorm / fused-add RMSNorm, SiLU*mul, RoPE (+ fused KV-pool store, a2/a3),
// row scatter / gather, kv_indices + positions builders (a4), greedy argmax (a9).
// All loads/stores are 16-byte vectors when the layout allows; math is fp32.
#include "common.h"

#include <stdarg.h>

namespace semipd {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------
// RMSNorm: one workgroup per row, the row lives in registers between the
// sum-of-squares pass and the scale pass (one HBM read, one write).
// ---------------------------------------------------------------------------
// QUANT: the normalised row is also quantised per group of 8 * lanes_per_group elements (per_token_group_quant_fp8
// of the output, layers/quantization/fp8_kernel.py:99-115) in the same pass: q [rows, hidden] e4m3fn, qs [rows,
// hidden / group].  The values quantised are the T-rounded outputs, so the bytes equal those of the separate call.
// PLANES: the input row is not in `in` but the sum, in slice order, of `n_planes` fp32 planes [rows, hidden]
// (the K slices of csrc/stream_linear.hip), rounded to T first -- exactly the row the GEMM's own reduction
// launch would have written, so the fusion changes no bit and saves that launch.
// Workgroup width of every form of the kernel (the plain, fused-add, plane-summing and quantising launches must agree: the
// order of the sum of squares, hence the bits, follows from it).  One 16-byte vector per lane up to hidden 8192: a row is
// one workgroup = one CU, and what bounds a decode-sized call is how many loads that CU has in flight (hidden 8192 with
// two planes: 6.4 us on 256 lanes x 4 vectors; SEMIPD_RMS_WIDE=0 restores that rule for A/B runs).
static int rms_threads(int nvec) {
  static const bool wide = [] { const char* e = getenv("SEMIPD_RMS_WIDE"); return !(e && atoi(e) == 0); }();
  if (!wide) return nvec <= 64 ? 64 : nvec <= 128 ? 128 : nvec <= 1024 ? 256 : 512;
  return nvec <= 64 ? 64 : nvec <= 128 ? 128 : nvec <= 256 ? 256 : nvec <= 512 ? 512 : nvec <= 2048 ? 1024 : 512;
}

template <typename T, int MAXV, bool FUSED, bool QUANT = false, bool PLANES = false>
__global__ void __launch_bounds__(MAXV <= 2 ? 1024 : 512)
rmsnorm_vec_kernel(T* __restrict__ out, T* __restrict__ in, T* __restrict__ res,
                   const T* __restrict__ w, int64_t in_stride, int64_t out_stride, int nvec,
                   int hidden, float eps, uint8_t* __restrict__ q = nullptr, float* __restrict__ qs = nullptr,
                   int lanes_per_group = 16, float q_eps = 1e-10f, const float* __restrict__ planes = nullptr,
                   int n_planes = 0, int64_t plane_elems = 0) {
  constexpr int V = Elem<T>::kVec;
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  T* in_row = PLANES ? nullptr : in + row * in_stride;
  T* out_row = out + row * out_stride;
  T* res_row = FUSED ? res + row * in_stride : nullptr;
  float x[MAXV][V];
  float ss = 0.f;
  // the weight row does not depend on the reduction: fetch it up front for rows short enough to keep it in
  // registers (decode-sized calls are latency-bound: one dependent global load less on the critical path)
  Vec16<T> wv[MAXV <= 2 ? MAXV : 1];
  if constexpr (MAXV <= 2) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = threadIdx.x + i * blockDim.x;
      if (v < nvec) wv[i] = load16(w + (int64_t)v * V);
    }
  }
  // PLANES: the plane sums of ALL of this thread's vectors first, every load of a batch of planes in flight at once
  // (planes_sum_f4, common.h); same summation order as before
  float psum[PLANES ? MAXV : 1][8];
  if constexpr (PLANES) {
    static_assert(!PLANES || Elem<T>::kVec == 8, "plane input: 16-bit activations");
    const float* pp[2 * MAXV];
    float4 acc4[2 * MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = min(threadIdx.x + i * blockDim.x, (unsigned)(nvec - 1));   // (tail threads re-read the last vector)
      pp[2 * i] = planes + row * (int64_t)hidden + (int64_t)v * V;
      pp[2 * i + 1] = pp[2 * i] + 4;
    }
    planes_sum_f4<2 * MAXV>(pp, n_planes, plane_elems, acc4);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      psum[i][0] = acc4[2 * i].x; psum[i][1] = acc4[2 * i].y; psum[i][2] = acc4[2 * i].z; psum[i][3] = acc4[2 * i].w;
      psum[i][4] = acc4[2 * i + 1].x; psum[i][5] = acc4[2 * i + 1].y; psum[i][6] = acc4[2 * i + 1].z; psum[i][7] = acc4[2 * i + 1].w;
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = threadIdx.x + i * blockDim.x;
    if (v < nvec) {
      Vec16<T> a;
      if constexpr (PLANES) {
#pragma unroll
        for (int j = 0; j < V; ++j) a.e[j] = Elem<T>::from_f(psum[i][j]);
      } else {
        a = load16(in_row + (int64_t)v * V);
      }
      if (FUSED) {
        Vec16<T> r = load16(res_row + (int64_t)v * V);
        Vec16<T> s;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          x[i][j] = Elem<T>::to_f(a.e[j]) + Elem<T>::to_f(r.e[j]);
          s.e[j] = Elem<T>::from_f(x[i][j]);
        }
        store16(res_row + (int64_t)v * V, s);
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) x[i][j] = Elem<T>::to_f(a.e[j]);
      }
#pragma unroll
      for (int j = 0; j < V; ++j) ss += x[i][j] * x[i][j];
    }
  }
  ss = block_sum(ss, red);
  const float rs = rsqrtf(ss / (float)hidden + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = threadIdx.x + i * blockDim.x;
    if (v < nvec) {
      Vec16<T> ww;
      if constexpr (MAXV <= 2) ww = wv[i];
      else ww = load16(w + (int64_t)v * V);
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.e[j] = Elem<T>::from_f(x[i][j] * rs * Elem<T>::to_f(ww.e[j]));
      store16(out_row + (int64_t)v * V, o);
      if constexpr (QUANT && V == 8) {
        // groups are aligned runs of lanes (hidden % group == 0): whole groups are inside or outside this branch
        float y[8], amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          y[j] = Elem<T>::to_f(o.e[j]);
          amax = fmaxf(amax, fabsf(y[j]));
        }
        for (int off = 1; off < lanes_per_group; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
        amax = fmaxf(amax, q_eps);
        const float y_s = amax / 448.0f;
        const float y_s_inv = 1.0f / y_s;
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = fminf(fmaxf(y[j] * y_s_inv, -448.0f), 448.0f);
        uint2 p;
        p.x = F8Cvt<f8e4m3_t>::pack2<false>(y[0], y[1], 0u);
        p.x = F8Cvt<f8e4m3_t>::pack2<true>(y[2], y[3], p.x);
        p.y = F8Cvt<f8e4m3_t>::pack2<false>(y[4], y[5], 0u);
        p.y = F8Cvt<f8e4m3_t>::pack2<true>(y[6], y[7], p.y);
        *reinterpret_cast<uint2*>(q + (row * nvec + v) * 8) = p;
        if ((v & (lanes_per_group - 1)) == 0) qs[row * (nvec / lanes_per_group) + v / lanes_per_group] = y_s;
      }
    }
  }
}

// Scalar fallback (hidden not a multiple of the vector width, or unaligned rows).
template <typename T, bool FUSED>
__global__ void rmsnorm_scalar_kernel(T* __restrict__ out, T* __restrict__ in, T* __restrict__ res,
                                      const T* __restrict__ w, int64_t in_stride,
                                      int64_t out_stride, int hidden, float eps) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  T* in_row = in + row * in_stride;
  T* out_row = out + row * out_stride;
  T* res_row = FUSED ? res + row * in_stride : nullptr;
  float ss = 0.f;
  for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
    float x = Elem<T>::to_f(in_row[i]);
    if (FUSED) {
      x += Elem<T>::to_f(res_row[i]);
    }
    ss += x * x;
  }
  ss = block_sum(ss, red);
  const float rs = rsqrtf(ss / (float)hidden + eps);
  for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
    float x = Elem<T>::to_f(in_row[i]);
    if (FUSED) {
      x += Elem<T>::to_f(res_row[i]);
      res_row[i] = Elem<T>::from_f(x);
    }
    out_row[i] = Elem<T>::from_f(x * rs * Elem<T>::to_f(w[i]));
  }
}

template <typename T, bool FUSED>
static int launch_rmsnorm(T* out, T* in, T* res, const T* w, int64_t T_rows, int64_t hidden,
                          int64_t in_stride, int64_t out_stride, float eps, hipStream_t st) {
  constexpr int V = Elem<T>::kVec;
  if (T_rows == 0) return 0;
  const bool vec_ok = (hidden % V == 0) && (in_stride % V == 0) && (out_stride % V == 0) &&
                      aligned16(out) && aligned16(in) && aligned16(w) && (!FUSED || aligned16(res));
  const int nvec = (int)(hidden / V);
  if (vec_ok && nvec <= 512 * 16) {
    int threads = rms_threads(nvec);
    // small rows: fewer threads, each with one vector
    int per = (nvec + threads - 1) / threads;
    dim3 grid((unsigned)T_rows), block(threads);
#define RMS_LAUNCH(MV)                                                                        \
  hipLaunchKernelGGL((rmsnorm_vec_kernel<T, MV, FUSED>), grid, block, 0, st, out, in, res, w, \
                     in_stride, out_stride, nvec, (int)hidden, eps)
    if (per <= 1) RMS_LAUNCH(1);
    else if (per <= 2) RMS_LAUNCH(2);
    else if (per <= 4) RMS_LAUNCH(4);
    else if (per <= 8) RMS_LAUNCH(8);
    else RMS_LAUNCH(16);
#undef RMS_LAUNCH
  } else {
    hipLaunchKernelGGL((rmsnorm_scalar_kernel<T, FUSED>), dim3((unsigned)T_rows), dim3(256), 0, st,
                       out, in, res, w, in_stride, out_stride, (int)hidden, eps);
  }
  return launch_status("rmsnorm");
}

// ---------------------------------------------------------------------------
// SiLU * mul
// ---------------------------------------------------------------------------
__device__ inline float silu_f(float x) { return x / (1.0f + __expf(-x)); }

template <typename T>
__global__ void silu_and_mul_vec_kernel(T* __restrict__ out, const T* __restrict__ in,
                                        int64_t num_tokens, int dvec) {
  constexpr int V = Elem<T>::kVec;
  const int64_t total = num_tokens * dvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / dvec;
    const int c = (int)(i - t * dvec);
    const T* row = in + t * (int64_t)dvec * 2 * V;
    Vec16<T> a = load16(row + (int64_t)c * V);
    Vec16<T> b = load16(row + (int64_t)(dvec + c) * V);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; ++j)
      o.e[j] = Elem<T>::from_f(silu_f(Elem<T>::to_f(a.e[j])) * Elem<T>::to_f(b.e[j]));
    store16(out + t * (int64_t)dvec * V + (int64_t)c * V, o);
  }
}

template <typename T>
__global__ void silu_and_mul_scalar_kernel(T* __restrict__ out, const T* __restrict__ in,
                                           int64_t num_tokens, int64_t d) {
  const int64_t total = num_tokens * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / d, c = i - t * d;
    const float a = Elem<T>::to_f(in[t * 2 * d + c]);
    const float b = Elem<T>::to_f(in[t * 2 * d + d + c]);
    out[i] = Elem<T>::from_f(silu_f(a) * b);
  }
}

// ---------------------------------------------------------------------------
// RoPE (in place) and fused RoPE + KV-pool store.
// Work item = one 16-byte vector of the first rotary half (neox) or of
// interleaved pairs (GPT-J) of one (token, head).
// ---------------------------------------------------------------------------
template <typename T, bool INTERLEAVE, typename KV = T>
__device__ inline void rope_item(T* __restrict__ head_ptr, KV* __restrict__ mirror_ptr,
                                 const float* __restrict__ cs, int rot_dim, int item) {
  // head_ptr: start of this head's row in place; mirror_ptr: optional second
  // destination (KV pool row) or nullptr.
  constexpr int V = Elem<T>::kVec;
  const int half = rot_dim >> 1;
  if (!INTERLEAVE) {
    const int i0 = item * V;  // pair index base, pairs (i, i+half)
    Vec16<T> a = load16(head_ptr + i0);
    Vec16<T> b = load16(head_ptr + half + i0);
    Vec16<T> oa, ob;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float c = cs[i0 + j], s = cs[half + i0 + j];
      const float x1 = Elem<T>::to_f(a.e[j]), x2 = Elem<T>::to_f(b.e[j]);
      float r1, r2;
      rope_pair(x1, x2, c, s, r1, r2);
      oa.e[j] = Elem<T>::from_f(r1);
      ob.e[j] = Elem<T>::from_f(r2);
    }
    store16(head_ptr + i0, oa);
    store16(head_ptr + half + i0, ob);
    if (mirror_ptr) {
      KVTraits<T, KV>::store8(mirror_ptr + i0, oa);
      KVTraits<T, KV>::store8(mirror_ptr + half + i0, ob);
    }
  } else {
    const int e0 = item * V;  // element base; pairs (e0+2j, e0+2j+1), pair index e0/2+j
    Vec16<T> a = load16(head_ptr + e0);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V / 2; ++j) {
      const int p = (e0 >> 1) + j;
      const float c = cs[p], s = cs[half + p];
      const float x1 = Elem<T>::to_f(a.e[2 * j]), x2 = Elem<T>::to_f(a.e[2 * j + 1]);
      float r1, r2;
      rope_pair(x1, x2, c, s, r1, r2);
      o.e[2 * j] = Elem<T>::from_f(r1);
      o.e[2 * j + 1] = Elem<T>::from_f(r2);
    }
    store16(head_ptr + e0, o);
    if (mirror_ptr) KVTraits<T, KV>::store8(mirror_ptr + e0, o);
  }
}

// one workgroup per token; handles q rotate, k rotate(+pool), k pass-through(+pool), v(+pool)
template <typename T, bool INTERLEAVE, bool STORE, typename KV = T>
__global__ void rope_vec_kernel(T* __restrict__ q, T* __restrict__ k, const T* __restrict__ v,
                                KV* __restrict__ k_buf, KV* __restrict__ v_buf,
                                const int64_t* __restrict__ loc, const float* __restrict__ cache,
                                const int64_t* __restrict__ positions, int Hq, int Hk, int head,
                                int vhead, int rot_dim, int64_t q_stride, int64_t k_stride,
                                int64_t v_stride, int64_t kbuf_stride, int64_t vbuf_stride) {
  constexpr int V = Elem<T>::kVec;
  const int64_t t = blockIdx.x;
  const int64_t pos = positions[t];
  const float* cs = cache + pos * rot_dim;
  const int items_per_head = INTERLEAVE ? rot_dim / V : (rot_dim / 2) / V;
  const int q_items = Hq * items_per_head;
  const int k_items = Hk * items_per_head;
  T* q_row = q + t * q_stride;
  T* k_row = k + t * k_stride;
  int64_t dst = 0;
  if (STORE) dst = loc[t];
  // ONE pass over every kind of work item of the token -- q rotations, k rotations (+ pool), k pass-through (+ pool), v rows
  // (+ pool) -- with a block as wide as the item count (launch_rope): every lane issues its loads at once and the token
  // costs one memory latency.  (Until round 5: three loops one after the other on 256 threads, i.e. up to five dependent
  // load -> store rounds per token; 20 us alone / 34 us next to a decode instance for a 1.3 k-token Llama-3-8B batch.)
  const int n_rot = q_items + k_items;
  const int pass_vec = STORE ? (head - rot_dim) / V : 0;
  const int n_pass = Hk * pass_vec;
  const int v_vec = STORE ? Hk * vhead / V : 0;
  const int total = n_rot + n_pass + v_vec;
  for (int it = threadIdx.x; it < total; it += blockDim.x) {
    if (it < q_items) {
      const int h = it / items_per_head, i = it - h * items_per_head;
      rope_item<T, INTERLEAVE, KV>(q_row + h * head, (KV*)nullptr, cs, rot_dim, i);
    } else if (it < n_rot) {
      const int kk = it - q_items;
      const int h = kk / items_per_head, i = kk - h * items_per_head;
      KV* mirror = STORE ? k_buf + dst * kbuf_stride + h * head : nullptr;
      rope_item<T, INTERLEAVE, KV>(k_row + h * head, mirror, cs, rot_dim, i);
    } else if constexpr (STORE) {
      if (it < n_rot + n_pass) {
        // pass-through part of k (rot_dim < head)
        const int pp = it - n_rot;
        const int h = pp / pass_vec, i = pp - h * pass_vec;
        Vec16<T> a = load16(k_row + h * head + rot_dim + i * V);
        KVTraits<T, KV>::store8(k_buf + dst * kbuf_stride + h * head + rot_dim + i * V, a);
      } else {
        const int vv = it - n_rot - n_pass;
        const T* v_row = v + t * v_stride;
        Vec16<T> a = load16(v_row + vv * V);
        KVTraits<T, KV>::store8(v_buf + dst * vbuf_stride + vv * V, a);
      }
    }
  }
}

// Decode-step form for a qkv row that is still K-slice planes of the streaming GEMM (stream_linear.hip): the fp32 planes
// [n_planes][tokens][Hq*head + 2*Hk*head] are summed in slice order and rounded to T -- the bits splitk_planes_reduce
// would have written -- then q is rotated into q_out, k rotated into the pool, v copied into the pool.  One launch
// instead of the reduction + rope_vec_kernel; neox pairing, rot_dim == head (Llama-shaped heads).
template <typename T, typename KV = T>
__global__ void rope_planes_kernel(T* __restrict__ q_out, const float* __restrict__ planes, int n_planes,
                                   int64_t plane_elems, int64_t row_elems, KV* __restrict__ k_buf, KV* __restrict__ v_buf,
                                   const int64_t* __restrict__ loc, const float* __restrict__ cache,
                                   const int64_t* __restrict__ positions, int Hq, int Hk, int head, int64_t q_stride,
                                   int64_t kbuf_stride, int64_t vbuf_stride) {
  constexpr int V = Elem<T>::kVec;
  static_assert(V == 8, "16-bit activations");
  const int64_t t = blockIdx.x;
  const float* cs = cache + positions[t] * head;
  const int64_t dst = loc[t];
  const int half = head >> 1, items_per_head = half / V;
  const float* row = planes + t * row_elems;
  auto sum8 = [&](int64_t col, float (&f)[8]) { planes_sum8(row + col, n_planes, plane_elems, f); };
  const int qk_items = (Hq + Hk) * items_per_head;
  const int v_vec = Hk * head / V;
  const int64_t v_col0 = (int64_t)(Hq + Hk) * head;
  // grid = (tokens, parts): the items of a token -- rotations of q and k, then the v vectors -- are independent, so a
  // decode batch of 32 tokens spreads over 32 x parts workgroups instead of 32 (a workgroup's plane reads, n_planes x
  // 24 KB per Llama-3-8B token, were bound by what ONE CU can pull), and a lane has one item: one memory latency.
  for (int it = blockIdx.y * blockDim.x + threadIdx.x; it < qk_items + v_vec; it += gridDim.y * blockDim.x) {
    if (it < qk_items) {
      const int h = it / items_per_head, i0 = (it - h * items_per_head) * V;
      float fa[8], fb[8];
      planes_sum8x2(row + (int64_t)h * head + i0, row + (int64_t)h * head + half + i0, n_planes, plane_elems, fa, fb);
      Vec16<T> oa, ob;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float c = cs[i0 + j], sn = cs[half + i0 + j];
        const float x1 = Elem<T>::to_f(Elem<T>::from_f(fa[j])), x2 = Elem<T>::to_f(Elem<T>::from_f(fb[j]));
        float r1, r2;
        rope_pair(x1, x2, c, sn, r1, r2);
        oa.e[j] = Elem<T>::from_f(r1);
        ob.e[j] = Elem<T>::from_f(r2);
      }
      if (h < Hq) {
        T* qh = q_out + t * q_stride + (int64_t)h * head;
        store16(qh + i0, oa);
        store16(qh + half + i0, ob);
      } else {
        KV* kh = k_buf + dst * kbuf_stride + (int64_t)(h - Hq) * head;
        KVTraits<T, KV>::store8(kh + i0, oa);
        KVTraits<T, KV>::store8(kh + half + i0, ob);
      }
    } else {
      const int vv = it - qk_items;
      float f[8];
      sum8(v_col0 + (int64_t)vv * V, f);
      Vec16<T> a;
#pragma unroll
      for (int j = 0; j < V; ++j) a.e[j] = Elem<T>::from_f(f[j]);
      KVTraits<T, KV>::store8(v_buf + dst * vbuf_stride + (int64_t)vv * V, a);
    }
  }
}

// MLA decode step, everything between the merged [q | kv_a] GEMM and the attention in ONE launch (one workgroup per token).
// The GEMM output is still the fp32 K-slice planes [n_planes][tokens][Hq * (nope + rope) + lora + rope]; every value is
// first summed in slice order and rounded to T -- the bits the GEMM's own reduction writes -- then
//   q_nope [t, h, 0:nope]                 -> q_nope_out (dense, the operand of the W_kc absorption)
//   q_pe   [t, h, nope:nope+rope]         -> RoPE (GPT-J pairs) -> q_input[t, h, lora : lora + rope]
//   latent [t, 0:lora]                    -> RMSNorm * kv_a_layernorm -> pool row loc[t], columns 0 : lora
//   k_pe   [t, lora:lora+rope]            -> RoPE -> pool row loc[t], columns lora : lora + rope
// i.e. the two reductions, ops.rmsnorm, apply_rope_strided_inplace, the q_pe copy into q_input and set_kv_buffer of
// DeepseekV2AttentionMLA.forward_absorb (models/deepseek_v2.py:633-706 in the reference): six launches.  The norm is the
// one-wave form of rmsnorm_vec_kernel (lora = 512: lane v owns vector v), the rotation is rope_item's: same bits.
template <typename T, typename KV = T>
__global__ void __launch_bounds__(256)
mla_decode_prep_kernel(T* __restrict__ q_nope_out, T* __restrict__ q_input, KV* __restrict__ kv_buf,
                       const float* __restrict__ planes, int n_planes, int64_t plane_elems, int64_t row_elems,
                       const int64_t* __restrict__ loc, const float* __restrict__ cache, const int64_t* __restrict__ positions,
                       const T* __restrict__ norm_w, float eps, int Hq, int nope, int rope, int lora, int64_t q_input_ts,
                       int64_t q_input_hs, int64_t kvbuf_stride, const T* __restrict__ q_src = nullptr, int64_t q_src_ts = 0,
                       int64_t q_src_hs = 0, const T* __restrict__ lat_src = nullptr, int64_t lat_src_ts = 0) {
  // n_planes == 0: the rows are already tensors of T -- q_src [tokens, Hq, nope + rope] and lat_src [tokens, lora + rope]
  // through their strides (the q_lora / block-fp8 path, whose GEMMs reduce themselves): same work minus the plane sums,
  // q_nope stays where it is (q_nope_out unused)
  constexpr int V = Elem<T>::kVec;
  static_assert(V == 8, "16-bit activations");
  const int64_t t = blockIdx.x;
  const float* row = planes + t * row_elems;
  const int qk = nope + rope;
  const int64_t kv_col0 = (int64_t)Hq * qk;
  KV* pool_row = kv_buf + loc[t] * kvbuf_stride;
  const float* cs = cache + positions[t] * rope;
  const int half = rope >> 1;
  const bool from_rows = n_planes == 0;
  auto sum8_T = [&](int64_t col) __attribute__((always_inline)) {
    if (from_rows) {
      if (col >= kv_col0) return load16(lat_src + t * lat_src_ts + (col - kv_col0));
      const int h = (int)(col / qk);
      return load16(q_src + t * q_src_ts + h * q_src_hs + (col - (int64_t)h * qk));
    }
    float f[8];
    planes_sum8(row + col, n_planes, plane_elems, f);
    Vec16<T> a;
#pragma unroll
    for (int j = 0; j < V; ++j) a.e[j] = Elem<T>::from_f(f[j]);
    return a;
  };
  // the rotation of rope_item<T, true> on a vector already in registers (element base e0 inside the rotary slice)
  auto rotate = [&](const Vec16<T>& a, int e0) __attribute__((always_inline)) {
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V / 2; ++j) {
      const int p = (e0 >> 1) + j;
      const float c = cs[p], s = cs[half + p];
      const float x1 = Elem<T>::to_f(a.e[2 * j]), x2 = Elem<T>::to_f(a.e[2 * j + 1]);
      float r1, r2;
      rope_pair(x1, x2, c, s, r1, r2);
      o.e[2 * j] = Elem<T>::from_f(r1);
      o.e[2 * j + 1] = Elem<T>::from_f(r2);
    }
    return o;
  };
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave == 0) {
    // latent: lane v owns vector v (lora / 8 <= 64 vectors), the one-wave rmsnorm_vec_kernel<T, 1, false>
    const int nvec = lora / V;
    float x[V];
    float ss = 0.f;
    if (lane < nvec) {
      const Vec16<T> a = sum8_T(kv_col0 + (int64_t)lane * V);
#pragma unroll
      for (int j = 0; j < V; ++j) x[j] = Elem<T>::to_f(a.e[j]);
#pragma unroll
      for (int j = 0; j < V; ++j) ss += x[j] * x[j];
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)lora + eps);
    if (lane < nvec) {
      const Vec16<T> ww = load16(norm_w + (int64_t)lane * V);
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.e[j] = Elem<T>::from_f(x[j] * rs * Elem<T>::to_f(ww.e[j]));
      KVTraits<T, KV>::store8(pool_row + (int64_t)lane * V, o);
    }
    // k_pe: rope / 8 vectors
    if (lane < rope / V) {
      const Vec16<T> a = sum8_T(kv_col0 + lora + (int64_t)lane * V);
      KVTraits<T, KV>::store8(pool_row + lora + (int64_t)lane * V, rotate(a, lane * V));
    }
    return;
  }
  // waves 1 .. 3: the q heads -- nope / 8 plain vectors and rope / 8 rotated vectors per head
  const int per_head = qk / V, nope_v = nope / V;
  for (int it = threadIdx.x - 64; it < Hq * per_head; it += blockDim.x - 64) {
    const int h = it / per_head, i = it - h * per_head;
    if (from_rows && i < nope_v) continue;
    const Vec16<T> a = sum8_T((int64_t)h * qk + (int64_t)i * V);
    if (i < nope_v) {
      store16(q_nope_out + (t * Hq + h) * (int64_t)nope + (int64_t)i * V, a);
    } else {
      const int e0 = (i - nope_v) * V;
      store16(q_input + t * q_input_ts + h * q_input_hs + lora + e0, rotate(a, e0));
    }
  }
}

// scalar fallback: any rot_dim (even), any alignment
template <typename T, bool STORE, typename KV = T>
__global__ void rope_scalar_kernel(T* __restrict__ q, T* __restrict__ k, const T* __restrict__ v,
                                   KV* __restrict__ k_buf, KV* __restrict__ v_buf,
                                   const int64_t* __restrict__ loc, const float* __restrict__ cache,
                                   const int64_t* __restrict__ positions, int Hq, int Hk, int head,
                                   int vhead, int rot_dim, int64_t q_stride, int64_t k_stride,
                                   int64_t v_stride, int64_t kbuf_stride, int64_t vbuf_stride,
                                   int interleave) {
  const int64_t t = blockIdx.x;
  const int64_t pos = positions[t];
  const float* cs = cache + pos * rot_dim;
  const int half = rot_dim >> 1;
  int64_t dst = 0;
  if (STORE) dst = loc[t];
  const int pairs = (Hq + Hk) * half;
  for (int it = threadIdx.x; it < pairs; it += blockDim.x) {
    const int h = it / half, p = it - h * half;
    T* row = h < Hq ? q + t * q_stride + h * head : k + t * k_stride + (h - Hq) * head;
    const int i1 = interleave ? 2 * p : p, i2 = interleave ? 2 * p + 1 : p + half;
    const float c = cs[p], s = cs[half + p];
    const float x1 = Elem<T>::to_f(row[i1]), x2 = Elem<T>::to_f(row[i2]);
    float r1, r2;
    rope_pair(x1, x2, c, s, r1, r2);
    const T o1 = Elem<T>::from_f(r1), o2 = Elem<T>::from_f(r2);
    row[i1] = o1;
    row[i2] = o2;
    if (STORE && h >= Hq) {
      KV* m = k_buf + dst * kbuf_stride + (h - Hq) * head;
      KVTraits<T, KV>::store1(m + i1, o1);
      KVTraits<T, KV>::store1(m + i2, o2);
    }
  }
  if (STORE) {
    const int pass = head - rot_dim;
    for (int it = threadIdx.x; it < Hk * pass; it += blockDim.x) {
      const int h = it / pass, i = it - h * pass;
      KVTraits<T, KV>::store1(k_buf + dst * kbuf_stride + h * head + rot_dim + i,
                              k[t * k_stride + h * head + rot_dim + i]);
    }
    for (int it = threadIdx.x; it < Hk * vhead; it += blockDim.x)
      KVTraits<T, KV>::store1(v_buf + dst * vbuf_stride + it, v[t * v_stride + it]);
  }
}

// RoPE on head slices that are not densely packed (MLA: q_pe = q[..., 128:], head stride 192;
// k_pe = latent[..., 512:], one "head" of 64 inside a 576-wide row).  q / k point at the first
// rotary element of head 0; strides in elements.
template <typename T, bool INTERLEAVE, bool VECT>
__global__ void rope_strided_kernel(T* __restrict__ q, T* __restrict__ k, const float* __restrict__ cache,
                                    const int64_t* __restrict__ positions, int Hq, int Hk, int rot_dim,
                                    int64_t q_ts, int64_t q_hs, int64_t k_ts, int64_t k_hs) {
  constexpr int V = Elem<T>::kVec;
  const int64_t t = blockIdx.x;
  const float* cs = cache + positions[t] * rot_dim;
  const int half = rot_dim >> 1;
  if (VECT) {
    const int iph = INTERLEAVE ? rot_dim / V : half / V;
    for (int it = threadIdx.x; it < (Hq + Hk) * iph; it += blockDim.x) {
      const int h = it / iph, i = it - h * iph;
      T* hp = h < Hq ? q + t * q_ts + h * q_hs : k + t * k_ts + (h - Hq) * k_hs;
      rope_item<T, INTERLEAVE, T>(hp, (T*)nullptr, cs, rot_dim, i);
    }
  } else {
    for (int it = threadIdx.x; it < (Hq + Hk) * half; it += blockDim.x) {
      const int h = it / half, p = it - h * half;
      T* hp = h < Hq ? q + t * q_ts + h * q_hs : k + t * k_ts + (h - Hq) * k_hs;
      const int i1 = INTERLEAVE ? 2 * p : p, i2 = INTERLEAVE ? 2 * p + 1 : p + half;
      const float c = cs[p], s = cs[half + p];
      const float x1 = Elem<T>::to_f(hp[i1]), x2 = Elem<T>::to_f(hp[i2]);
      float r1, r2;
      rope_pair(x1, x2, c, s, r1, r2);
      hp[i1] = Elem<T>::from_f(r1);
      hp[i2] = Elem<T>::from_f(r2);
    }
  }
}

template <typename T>
static int launch_rope_strided(T* q, T* k, const float* cache, const int64_t* positions, int64_t num_tokens,
                               int Hq, int Hk, int rot_dim, int64_t q_ts, int64_t q_hs, int64_t k_ts,
                               int64_t k_hs, int interleave, hipStream_t st) {
  constexpr int V = Elem<T>::kVec;
  const bool vec = aligned16(q) && aligned16(k) && q_ts % V == 0 && q_hs % V == 0 && k_ts % V == 0 &&
                   k_hs % V == 0 && (interleave ? rot_dim % V == 0 : (rot_dim / 2) % V == 0);
  dim3 grid((unsigned)num_tokens), block(128);
  if (interleave) {
    if (vec) hipLaunchKernelGGL((rope_strided_kernel<T, true, true>), grid, block, 0, st, q, k, cache, positions, Hq, Hk, rot_dim, q_ts, q_hs, k_ts, k_hs);
    else hipLaunchKernelGGL((rope_strided_kernel<T, true, false>), grid, block, 0, st, q, k, cache, positions, Hq, Hk, rot_dim, q_ts, q_hs, k_ts, k_hs);
  } else {
    if (vec) hipLaunchKernelGGL((rope_strided_kernel<T, false, true>), grid, block, 0, st, q, k, cache, positions, Hq, Hk, rot_dim, q_ts, q_hs, k_ts, k_hs);
    else hipLaunchKernelGGL((rope_strided_kernel<T, false, false>), grid, block, 0, st, q, k, cache, positions, Hq, Hk, rot_dim, q_ts, q_hs, k_ts, k_hs);
  }
  return launch_status("rope_strided");
}

template <typename T, bool STORE, typename KV = T>
static int launch_rope(T* q, T* k, const T* v, KV* k_buf, KV* v_buf, const int64_t* loc,
                       const float* cache, const int64_t* positions, int64_t num_tokens, int Hq,
                       int Hk, int head, int vhead, int rot_dim, int64_t q_stride, int64_t k_stride,
                       int64_t v_stride, int64_t kbuf_stride, int64_t vbuf_stride, int interleave,
                       hipStream_t st) {
  constexpr int V = Elem<T>::kVec;
  if (num_tokens == 0) return 0;
  bool vec_ok = aligned16(q) && aligned16(k) && (q_stride % V == 0) && (k_stride % V == 0) &&
                (head % V == 0) && (interleave ? rot_dim % V == 0 : (rot_dim / 2) % V == 0);
  if (STORE)  // pool rows: 8 elements per vector store = 16 bytes, or 8 bytes for fp8 rows
    vec_ok = vec_ok && aligned16(v) && aligned16(k_buf) && aligned16(v_buf) && (v_stride % V == 0) &&
             (kbuf_stride % 16 == 0) && (vbuf_stride % 16 == 0) && ((head - rot_dim) % V == 0) &&
             (vhead % V == 0) && (KVTraits<T, KV>::kF8 ? V == 8 : true);
  dim3 grid((unsigned)num_tokens), block(256);
  if (vec_ok) {
    // one lane per work item of a token, whole waves, at most 1024 (rope_vec_kernel walks what is left over)
    const int per_head = interleave ? rot_dim / V : (rot_dim / 2) / V;
    const int items = (Hq + Hk) * per_head + (STORE ? Hk * ((head - rot_dim) / V) + Hk * vhead / V : 0);
    block = dim3((unsigned)std::min(1024, std::max(64, (items + 63) / 64 * 64)));
    if (interleave)
      hipLaunchKernelGGL((rope_vec_kernel<T, true, STORE, KV>), grid, block, 0, st, q, k, v, k_buf, v_buf,
                         loc, cache, positions, Hq, Hk, head, vhead, rot_dim, q_stride, k_stride,
                         v_stride, kbuf_stride, vbuf_stride);
    else
      hipLaunchKernelGGL((rope_vec_kernel<T, false, STORE, KV>), grid, block, 0, st, q, k, v, k_buf,
                         v_buf, loc, cache, positions, Hq, Hk, head, vhead, rot_dim, q_stride,
                         k_stride, v_stride, kbuf_stride, vbuf_stride);
  } else {
    hipLaunchKernelGGL((rope_scalar_kernel<T, STORE, KV>), grid, block, 0, st, q, k, v, k_buf, v_buf, loc,
                       cache, positions, Hq, Hk, head, vhead, rot_dim, q_stride, k_stride, v_stride,
                       kbuf_stride, vbuf_stride, interleave);
  }
  return launch_status("rope");
}

// ---------------------------------------------------------------------------
// row scatter (dst indexed) / gather (src indexed); bytes, 16-byte vectors
// ---------------------------------------------------------------------------
template <bool SCATTER, int W>
__global__ void row_copy_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src,
                                const int64_t* __restrict__ index, int64_t rows, int64_t row_units,
                                int64_t dst_stride, int64_t src_stride) {
  // W = bytes per unit (16, 4, 2 or 1)
  const int64_t total = rows * row_units;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / row_units, c = i - r * row_units;
    const int64_t ix = index[r];
    const uint8_t* s = src + (SCATTER ? r : ix) * src_stride + c * W;
    uint8_t* d = dst + (SCATTER ? ix : r) * dst_stride + c * W;
    if (W == 16) *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
    else if (W == 4) *reinterpret_cast<uint32_t*>(d) = *reinterpret_cast<const uint32_t*>(s);
    else if (W == 2) *reinterpret_cast<uint16_t*>(d) = *reinterpret_cast<const uint16_t*>(s);
    else *d = *s;
  }
}

template <bool SCATTER>
static int launch_row_copy(void* dst, const void* src, const int64_t* index, int64_t rows,
                           int64_t row_bytes, int64_t dst_stride, int64_t src_stride,
                           hipStream_t st) {
  if (rows == 0 || row_bytes == 0) return 0;
  int W = 1;
  auto ok = [&](int w) {
    return row_bytes % w == 0 && dst_stride % w == 0 && src_stride % w == 0 &&
           (reinterpret_cast<uintptr_t>(dst) % w) == 0 && (reinterpret_cast<uintptr_t>(src) % w) == 0;
  };
  if (ok(16)) W = 16; else if (ok(4)) W = 4; else if (ok(2)) W = 2;
  const int64_t units = row_bytes / W;
  const int64_t total = rows * units;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  uint8_t* d = (uint8_t*)dst;
  const uint8_t* s = (const uint8_t*)src;
#define RC(WW) hipLaunchKernelGGL((row_copy_kernel<SCATTER, WW>), dim3(blocks), dim3(256), 0, st, d, s, index, rows, units, dst_stride, src_stride)
  if (W == 16) RC(16); else if (W == 4) RC(4); else if (W == 2) RC(2); else RC(1);
#undef RC
  return launch_status("row_copy");
}

// ---------------------------------------------------------------------------
// kv_indptr / kv_indices, positions
// ---------------------------------------------------------------------------
template <typename LT>
__global__ void build_kv_indices_kernel(const int32_t* __restrict__ req_to_token, int64_t stride,
                                        const int64_t* __restrict__ req_pool_indices,
                                        const LT* __restrict__ lens,
                                        const int32_t* __restrict__ start,
                                        int32_t* __restrict__ kv_indptr,
                                        int32_t* __restrict__ kv_indices, int64_t batch) {
  __shared__ float redf[16];
  __shared__ int red[16];
  (void)redf;
  const int b = blockIdx.x;
  // exclusive prefix of lens[0..b)
  int part = 0;
  for (int i = threadIdx.x; i < b; i += blockDim.x) part += (int)lens[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = part;
  __syncthreads();
  int off = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) off += red[i];
  const int len = (int)lens[b];
  if (threadIdx.x == 0) {
    if (b == 0) kv_indptr[0] = 0;
    kv_indptr[b + 1] = off + len;
  }
  const int s0 = start ? start[b] : 0;
  const int32_t* src = req_to_token + req_pool_indices[b] * stride + s0;
  for (int i = threadIdx.x; i < len; i += blockDim.x) kv_indices[off + i] = src[i];
}

__global__ void compute_positions_kernel(const int32_t* __restrict__ prefix_lens,
                                         const int32_t* __restrict__ extend_lens,
                                         int64_t* __restrict__ positions,
                                         int32_t* __restrict__ extend_start_loc, int64_t batch) {
  __shared__ int red[16];
  const int b = blockIdx.x;
  int part = 0;
  for (int i = threadIdx.x; i < b; i += blockDim.x) part += extend_lens[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = part;
  __syncthreads();
  int off = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) off += red[i];
  if (threadIdx.x == 0 && extend_start_loc) extend_start_loc[b] = off;
  const int len = extend_lens[b], pre = prefix_lens[b];
  for (int i = threadIdx.x; i < len; i += blockDim.x) positions[off + i] = (int64_t)(pre + i);
}

// ---------------------------------------------------------------------------
// greedy argmax: one workgroup per row; ties -> lowest index
// ---------------------------------------------------------------------------
__device__ inline void argmax_combine(float& bv, int64_t& bi, float v, int64_t i) {
  if (v > bv || (v == bv && i < bi)) {
    bv = v;
    bi = i;
  }
}

template <typename T>
__global__ void __launch_bounds__(1024)
argmax_kernel(const T* __restrict__ logits, void* __restrict__ out, int64_t vocab, int64_t stride,
              int out_is_i64) {
  constexpr int V = Elem<T>::kVec;
  __shared__ float sv[16];
  __shared__ int64_t si[16];
  const int64_t row = blockIdx.x;
  const T* p = logits + row * stride;
  float bv = -INFINITY;
  int64_t bi = INT64_MAX;
  const bool vec_ok = (stride % V == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15u) == 0);
  const int64_t nvec = vec_ok ? vocab / V : 0;
  for (int64_t v = threadIdx.x; v < nvec; v += blockDim.x) {
    Vec16<T> a = load16(p + v * V);
#pragma unroll
    for (int j = 0; j < V; ++j) argmax_combine(bv, bi, Elem<T>::to_f(a.e[j]), v * V + j);
  }
  for (int64_t i = nvec * V + threadIdx.x; i < vocab; i += blockDim.x)
    argmax_combine(bv, bi, Elem<T>::to_f(p[i]), i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int64_t oi = __shfl_xor(bi, o, 64);
    argmax_combine(bv, bi, ov, oi);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    sv[wid] = bv;
    si[wid] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) argmax_combine(bv, bi, sv[i], si[i]);
    if (bi == INT64_MAX) bi = 0;  // all -inf / NaN row
    if (out_is_i64) reinterpret_cast<int64_t*>(out)[row] = bi;
    else reinterpret_cast<int32_t*>(out)[row] = (int32_t)bi;
  }
}

// moe_sum: out[t,:] = sum_j in[t,j,:]; the fused form continues in registers with the two element-wise ops that follow
// it in DeepseekV2MoE.forward (models/deepseek_v2.py:139-160): * routed_scaling_factor, + shared_output -- each rounded to T
// like the separate kernels round it, so the bits are those of the three-launch sequence
// add_planes: the addend is still the fp32 K-slice planes [n_planes][tokens][hidden] of the GEMM that produces it (the shared
// experts' down_proj on the weight-streaming kernel): summed in slice order and rounded to T here -- that GEMM's reduction
template <typename T>
__global__ void moe_sum_kernel(T* __restrict__ out, const T* __restrict__ in, int64_t num_tokens,
                               int topk, int hvec, float scale = 1.f, const T* __restrict__ addend = nullptr,
                               int apply_scale = 0, const float* __restrict__ add_planes = nullptr, int n_planes = 0,
                               int64_t plane_elems = 0) {
  constexpr int V = Elem<T>::kVec;
  const int64_t total = num_tokens * hvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / hvec;
    const int c = (int)(i - t * hvec);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int k = 0; k < topk; ++k) {
      Vec16<T> a = load16(in + ((t * topk + k) * hvec + c) * V);
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += Elem<T>::to_f(a.e[j]);
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.e[j] = Elem<T>::from_f(acc[j]);
    if (apply_scale) {
#pragma unroll
      for (int j = 0; j < V; ++j) o.e[j] = Elem<T>::from_f(Elem<T>::to_f(o.e[j]) * scale);
    }
    if (addend != nullptr) {
      const Vec16<T> b = load16(addend + (t * hvec + c) * V);
#pragma unroll
      for (int j = 0; j < V; ++j) o.e[j] = Elem<T>::from_f(Elem<T>::to_f(o.e[j]) + Elem<T>::to_f(b.e[j]));
    }
    if constexpr (V == 8) {
      if (add_planes != nullptr) {
        float f[8];
        planes_sum8(add_planes + (t * hvec + c) * V, n_planes, plane_elems, f);
#pragma unroll
        for (int j = 0; j < V; ++j)
          o.e[j] = Elem<T>::from_f(Elem<T>::to_f(o.e[j]) + Elem<T>::to_f(Elem<T>::from_f(f[j])));
      }
    }
    store16(out + (t * hvec + c) * V, o);
  }
}

// ---- KV rows of the activation type -> fp8 pool rows (scatter by loc) ----------------------------
template <typename T, typename KV>
__global__ void kv_store_cvt_kernel(KV* __restrict__ buf, const T* __restrict__ src, const int64_t* __restrict__ loc,
                                    int64_t num_tokens, int64_t row_elems, int64_t buf_stride, int64_t src_stride,
                                    int vec) {
  const int64_t per_row = vec ? row_elems / 8 : row_elems;
  const int64_t total = num_tokens * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / per_row, c = i - t * per_row;
    const int64_t dst = loc[t];
    if (vec) KVTraits<T, KV>::store8(buf + dst * buf_stride + c * 8, load16(src + t * src_stride + c * 8));
    else KVTraits<T, KV>::store1(buf + dst * buf_stride + c, src[t * src_stride + c]);
  }
}

template <typename T, typename KV>
static int launch_kv_store_cvt(void* buf, const void* src, const int64_t* loc, int64_t num_tokens, int64_t row_elems,
                               int64_t buf_stride, int64_t src_stride, hipStream_t st) {
  const int vec = (row_elems % 8 == 0 && buf_stride % 8 == 0 && src_stride % 8 == 0 && aligned16(src) &&
                   (reinterpret_cast<uintptr_t>(buf) & 7u) == 0) ? 1 : 0;
  const int64_t total = num_tokens * (vec ? row_elems / 8 : row_elems);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL((kv_store_cvt_kernel<T, KV>), dim3(blocks), dim3(256), 0, st, (KV*)buf, (const T*)src, loc,
                     num_tokens, row_elems, buf_stride, src_stride, vec);
  return launch_status("kv_store_cvt");
}

// fp8 pool rows are only defined for 16-bit activations
template <typename T>
static int kv_store_cvt_dispatch(void* buf, const void* src, const int64_t* loc, int64_t num_tokens,
                                 int64_t row_elems, int64_t buf_stride, int64_t src_stride, int kv_dtype,
                                 hipStream_t st) {
  if constexpr (Elem<T>::kVec == 8) {
    if (kv_dtype == SEMIPD_F8E5M2)
      return launch_kv_store_cvt<T, f8e5m2_t>(buf, src, loc, num_tokens, row_elems, buf_stride, src_stride, st);
    if (kv_dtype == SEMIPD_F8E4M3)
      return launch_kv_store_cvt<T, f8e4m3_t>(buf, src, loc, num_tokens, row_elems, buf_stride, src_stride, st);
  }
  set_error("kv_store_cvt: unsupported kv_dtype %d for this activation type", kv_dtype);
  return SEMIPD_EDTYPE;
}

template <typename T>
static int rope_kv_store_dispatch(void* q, void* k, const void* v, void* k_buf, void* v_buf, const int64_t* loc,
                                  const float* cache, const int64_t* positions, int64_t num_tokens, int Hq, int Hk,
                                  int head, int vhead, int rot_dim, int64_t q_stride, int64_t k_stride,
                                  int64_t v_stride, int64_t kbuf_stride, int64_t vbuf_stride, int interleave,
                                  int dtype, int kv_dtype, hipStream_t st) {
  if (kv_dtype == dtype)
    return launch_rope<T, true, T>((T*)q, (T*)k, (const T*)v, (T*)k_buf, (T*)v_buf, loc, cache, positions, num_tokens,
                                   Hq, Hk, head, vhead, rot_dim, q_stride, k_stride, v_stride, kbuf_stride,
                                   vbuf_stride, interleave, st);
  if constexpr (Elem<T>::kVec == 8) {
    if (kv_dtype == SEMIPD_F8E5M2)
      return launch_rope<T, true, f8e5m2_t>((T*)q, (T*)k, (const T*)v, (f8e5m2_t*)k_buf, (f8e5m2_t*)v_buf, loc, cache,
                                            positions, num_tokens, Hq, Hk, head, vhead, rot_dim, q_stride, k_stride,
                                            v_stride, kbuf_stride, vbuf_stride, interleave, st);
    if (kv_dtype == SEMIPD_F8E4M3)
      return launch_rope<T, true, f8e4m3_t>((T*)q, (T*)k, (const T*)v, (f8e4m3_t*)k_buf, (f8e4m3_t*)v_buf, loc, cache,
                                            positions, num_tokens, Hq, Hk, head, vhead, rot_dim, q_stride, k_stride,
                                            v_stride, kbuf_stride, vbuf_stride, interleave, st);
  }
  set_error("rope_kv_store: unsupported kv_dtype %d for activation dtype %d", kv_dtype, dtype);
  return SEMIPD_EDTYPE;
}

}  // namespace semipd

using namespace semipd;

extern "C" {

int semipd_version(void) { return 100; }
const char* semipd_last_error(void) { return g_err; }

int semipd_rmsnorm(void* out, const void* in, const void* weight, int64_t num_tokens,
                   int64_t hidden, int64_t in_stride, int64_t out_stride, float eps, int dtype,
                   void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && hidden > 0, SEMIPD_EINVAL, "rmsnorm: bad sizes");
  SEMIPD_CHECK_ARG(num_tokens == 0 || (out && in && weight), SEMIPD_EINVAL, "rmsnorm: null pointer");
  SEMIPD_CHECK_ARG(num_tokens < (1ll << 31), SEMIPD_EINVAL, "rmsnorm: too many rows");
  SEMIPD_DISPATCH_DTYPE(dtype, T, return (launch_rmsnorm<T, false>((T*)out, (T*)in, nullptr, (const T*)weight, num_tokens, hidden, in_stride, out_stride, eps, as_stream(stream))));
  return 0;
}

int semipd_fused_add_rmsnorm(void* inout, void* residual, const void* weight, int64_t num_tokens,
                             int64_t hidden, float eps, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && hidden > 0, SEMIPD_EINVAL, "fused_add_rmsnorm: bad sizes");
  SEMIPD_CHECK_ARG(num_tokens == 0 || (inout && residual && weight), SEMIPD_EINVAL,
                   "fused_add_rmsnorm: null pointer");
  SEMIPD_CHECK_ARG(num_tokens < (1ll << 31), SEMIPD_EINVAL, "fused_add_rmsnorm: too many rows");
  SEMIPD_DISPATCH_DTYPE(dtype, T, return (launch_rmsnorm<T, true>((T*)inout, (T*)inout, (T*)residual, (const T*)weight, num_tokens, hidden, hidden, hidden, eps, as_stream(stream))));
  return 0;
}

int semipd_fused_add_rmsnorm_planes(void* out, void* residual, const void* weight, const float* planes, int n_planes,
                                    int64_t plane_elems, int64_t num_tokens, int64_t hidden, float eps, int dtype,
                                    void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && hidden > 0 && n_planes >= 1 && n_planes <= 64, SEMIPD_EINVAL,
                   "fused_add_rmsnorm_planes: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(out && residual && weight && planes, SEMIPD_EINVAL, "fused_add_rmsnorm_planes: null pointer");
  SEMIPD_CHECK_ARG(dtype == SEMIPD_BF16 || dtype == SEMIPD_F16, SEMIPD_EDTYPE, "fused_add_rmsnorm_planes: bf16 / f16 only");
  SEMIPD_CHECK_ARG(hidden % 8 == 0 && hidden <= 8 * 512 * 16 && aligned16(out) && aligned16(residual) && aligned16(weight) &&
                   aligned16(planes) && plane_elems % 4 == 0 && plane_elems >= num_tokens * hidden,
                   SEMIPD_EALIGN, "fused_add_rmsnorm_planes: hidden %% 8, 16-byte aligned rows required");
  const int nvec = (int)(hidden / 8);
  // the launch shape of launch_rmsnorm: same reduction order, hence the same bits as the unfused pair of launches
  const int threads = rms_threads(nvec);
  const int per = (nvec + threads - 1) / threads;
  dim3 grid((unsigned)num_tokens), block(threads);
#define RMSP_LAUNCH(MV)                                                                                              \
  hipLaunchKernelGGL((rmsnorm_vec_kernel<T, MV, true, false, true>), grid, block, 0, as_stream(stream), (T*)out,      \
                     (T*)nullptr, (T*)residual, (const T*)weight, hidden, hidden, nvec, (int)hidden, eps,             \
                     (uint8_t*)nullptr, (float*)nullptr, 16, 1e-10f, planes, n_planes, plane_elems)
  SEMIPD_DISPATCH_HALF(dtype, T, {
    if (per <= 1) RMSP_LAUNCH(1);
    else if (per <= 2) RMSP_LAUNCH(2);
    else if (per <= 4) RMSP_LAUNCH(4);
    else if (per <= 8) RMSP_LAUNCH(8);
    else RMSP_LAUNCH(16);
  });
#undef RMSP_LAUNCH
  return launch_status("fused_add_rmsnorm_planes");
}

int semipd_fused_add_rmsnorm_quant_fp8(void* inout, void* residual, const void* weight, void* q, float* qs,
                                       int64_t num_tokens, int64_t hidden, float eps, int group_size, float q_eps,
                                       int dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && hidden > 0 && num_tokens < (1ll << 31), SEMIPD_EINVAL,
                   "fused_add_rmsnorm_quant_fp8: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(inout && residual && weight && q && qs, SEMIPD_EINVAL, "fused_add_rmsnorm_quant_fp8: null pointer");
  SEMIPD_CHECK_ARG(group_size == 64 || group_size == 128 || group_size == 256 || group_size == 512, SEMIPD_ESHAPE,
                   "fused_add_rmsnorm_quant_fp8: group size %d is not one of 64, 128, 256, 512", group_size);
  SEMIPD_CHECK_ARG(hidden % group_size == 0 && hidden <= 8 * 512 * 2, SEMIPD_ESHAPE,
                   "fused_add_rmsnorm_quant_fp8: hidden size %lld must be a multiple of the group size and at most 8192",
                   (long long)hidden);
  SEMIPD_CHECK_ARG(dtype == SEMIPD_BF16 || dtype == SEMIPD_F16, SEMIPD_EDTYPE,
                   "fused_add_rmsnorm_quant_fp8: bf16 / f16 only");
  SEMIPD_CHECK_ARG(aligned16(inout) && aligned16(residual) && aligned16(weight) &&
                       (reinterpret_cast<uintptr_t>(q) & 7u) == 0,
                   SEMIPD_EALIGN, "fused_add_rmsnorm_quant_fp8: unaligned pointer");
  const int nvec = (int)(hidden / 8), lpg = group_size / 8;
  // the launch shape of launch_rmsnorm, so that the reduction order — hence every output bit — is the same as in
  // the unfused kernel
  const int threads = rms_threads(nvec);
  const int per = (nvec + threads - 1) / threads;
  dim3 grid((unsigned)num_tokens), block(threads);
  hipStream_t st = as_stream(stream);
#define RQ(T, MV)                                                                                             \
  hipLaunchKernelGGL((rmsnorm_vec_kernel<T, MV, true, true>), grid, block, 0, st, (T*)inout, (T*)inout,       \
                     (T*)residual, (const T*)weight, hidden, hidden, nvec, (int)hidden, eps, (uint8_t*)q, qs, \
                     lpg, q_eps)
  if (dtype == SEMIPD_BF16) {
    if (per <= 1) RQ(bf16_t, 1);
    else if (per <= 2) RQ(bf16_t, 2);
    else RQ(bf16_t, 4);
  } else {
    if (per <= 1) RQ(f16_t, 1);
    else if (per <= 2) RQ(f16_t, 2);
    else RQ(f16_t, 4);
  }
#undef RQ
  return launch_status("fused_add_rmsnorm_quant_fp8");
}

int semipd_silu_and_mul(void* out, const void* in, int64_t num_tokens, int64_t d, int dtype,
                        void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && d > 0, SEMIPD_EINVAL, "silu_and_mul: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(out && in, SEMIPD_EINVAL, "silu_and_mul: null pointer");
  hipStream_t st = as_stream(stream);
  SEMIPD_DISPATCH_DTYPE(dtype, T, {
    constexpr int V = Elem<T>::kVec;
    const int64_t total = num_tokens * d;
    if (d % V == 0 && aligned16(out) && aligned16(in)) {
      int64_t nv = total / V;
      int blocks = (int)((nv + 255) / 256);
      if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL((silu_and_mul_vec_kernel<T>), dim3(blocks), dim3(256), 0, st, (T*)out,
                         (const T*)in, num_tokens, (int)(d / V));
    } else {
      int blocks = (int)((total + 255) / 256);
      if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL((silu_and_mul_scalar_kernel<T>), dim3(blocks), dim3(256), 0, st, (T*)out,
                         (const T*)in, num_tokens, d);
    }
  });
  return launch_status("silu_and_mul");
}

int semipd_rope_inplace(void* q, void* k, const float* cos_sin_cache, const int64_t* positions,
                        int64_t num_tokens, int num_q_heads, int num_k_heads, int head_size,
                        int rot_dim, int64_t q_stride, int64_t k_stride, int interleave, int dtype,
                        void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && head_size > 0 && rot_dim > 0 && rot_dim <= head_size &&
                       (rot_dim % 2) == 0 && num_q_heads >= 0 && num_k_heads >= 0,
                   SEMIPD_EINVAL, "rope: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(q && k && cos_sin_cache && positions, SEMIPD_EINVAL, "rope: null pointer");
  SEMIPD_DISPATCH_DTYPE(dtype, T, return (launch_rope<T, false, T>((T*)q, (T*)k, (const T*)nullptr, (T*)nullptr, (T*)nullptr, nullptr, cos_sin_cache, positions, num_tokens, num_q_heads, num_k_heads, head_size, 0, rot_dim, q_stride, k_stride, 0, 0, 0, interleave, as_stream(stream))));
  return 0;
}

int semipd_rope_inplace_strided(void* q, void* k, const float* cos_sin_cache, const int64_t* positions,
                                int64_t num_tokens, int num_q_heads, int num_k_heads, int rot_dim,
                                int64_t q_token_stride, int64_t q_head_stride, int64_t k_token_stride,
                                int64_t k_head_stride, int interleave, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && rot_dim > 0 && (rot_dim % 2) == 0 && num_q_heads >= 0 && num_k_heads >= 0,
                   SEMIPD_EINVAL, "rope_strided: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(q && k && cos_sin_cache && positions, SEMIPD_EINVAL, "rope_strided: null pointer");
  SEMIPD_DISPATCH_DTYPE(dtype, T, return (launch_rope_strided<T>((T*)q, (T*)k, cos_sin_cache, positions, num_tokens, num_q_heads, num_k_heads, rot_dim, q_token_stride, q_head_stride, k_token_stride, k_head_stride, interleave, as_stream(stream))));
  return 0;
}

int semipd_rope_kv_store(void* q, void* k, const void* v, void* k_buf, void* v_buf,
                         const int64_t* loc, const float* cos_sin_cache, const int64_t* positions,
                         int64_t num_tokens, int num_q_heads, int num_k_heads, int head_size,
                         int v_head_size, int rot_dim, int64_t q_stride, int64_t k_stride,
                         int64_t v_stride, int64_t kbuf_stride, int64_t vbuf_stride, int interleave,
                         int dtype, int kv_dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && head_size > 0 && rot_dim > 0 && rot_dim <= head_size &&
                       (rot_dim % 2) == 0 && v_head_size > 0,
                   SEMIPD_EINVAL, "rope_kv_store: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(q && k && v && k_buf && v_buf && loc && cos_sin_cache && positions, SEMIPD_EINVAL,
                   "rope_kv_store: null pointer");
  SEMIPD_DISPATCH_DTYPE(dtype, T, return (rope_kv_store_dispatch<T>(q, k, v, k_buf, v_buf, loc, cos_sin_cache, positions, num_tokens, num_q_heads, num_k_heads, head_size, v_head_size, rot_dim, q_stride, k_stride, v_stride, kbuf_stride, vbuf_stride, interleave, dtype, kv_dtype, as_stream(stream))));
  return 0;
}

int semipd_rope_kv_store_planes(void* q_out, const float* planes, int n_planes, int64_t plane_elems, void* k_buf,
                                void* v_buf, const int64_t* loc, const float* cos_sin_cache, const int64_t* positions,
                                int64_t num_tokens, int num_q_heads, int num_k_heads, int head_size, int64_t q_stride,
                                int64_t kbuf_stride, int64_t vbuf_stride, int dtype, int kv_dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && head_size > 0 && n_planes >= 1 && num_q_heads > 0 && num_k_heads > 0, SEMIPD_EINVAL,
                   "rope_kv_store_planes: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(q_out && planes && k_buf && v_buf && loc && cos_sin_cache && positions, SEMIPD_EINVAL,
                   "rope_kv_store_planes: null pointer");
  const int64_t row_elems = (int64_t)(num_q_heads + 2 * num_k_heads) * head_size;
  SEMIPD_CHECK_ARG(head_size % 16 == 0 && q_stride % 8 == 0 && kbuf_stride % 16 == 0 && vbuf_stride % 16 == 0 &&
                   plane_elems % 4 == 0 && plane_elems >= num_tokens * row_elems && aligned16(q_out) && aligned16(planes) &&
                   aligned16(k_buf) && aligned16(v_buf), SEMIPD_EALIGN,
                   "rope_kv_store_planes: head_size %% 16, 16-byte aligned rows required");
  SEMIPD_CHECK_ARG(dtype == SEMIPD_BF16 || dtype == SEMIPD_F16, SEMIPD_EDTYPE, "rope_kv_store_planes: bf16 / f16 activations");
  hipStream_t st = as_stream(stream);
  // items of one token: (Hq + Hk) * head / 16 rotations + Hk * head / 8 v vectors; workgroups of 64..256 lanes, about 256
  // of them for a decode-sized batch (rope_planes_kernel)
  const int items = (num_q_heads + num_k_heads) * (head_size / 16) + num_k_heads * head_size / 8;
  int parts = (int)std::max<int64_t>(1, std::min<int64_t>((items + 63) / 64, 256 / num_tokens));
  int per_part = (items + parts - 1) / parts;
  int threads = std::min(256, (per_part + 63) / 64 * 64);
  dim3 grid((unsigned)num_tokens, (unsigned)parts), block((unsigned)threads);
#define RPK(TT, KVT)                                                                                                  \
  hipLaunchKernelGGL((rope_planes_kernel<TT, KVT>), grid, block, 0, st, (TT*)q_out, planes, n_planes, plane_elems,      \
                     row_elems, (KVT*)k_buf, (KVT*)v_buf, loc, cos_sin_cache, positions, num_q_heads, num_k_heads,     \
                     head_size, q_stride, kbuf_stride, vbuf_stride)
  SEMIPD_DISPATCH_HALF(dtype, T, {
    if (kv_dtype == dtype) RPK(T, T);
    else if (kv_dtype == SEMIPD_F8E5M2) RPK(T, f8e5m2_t);
    else if (kv_dtype == SEMIPD_F8E4M3) RPK(T, f8e4m3_t);
    else {
      set_error("rope_kv_store_planes: unsupported kv_dtype %d", kv_dtype);
      return SEMIPD_EDTYPE;
    }
  });
#undef RPK
  return launch_status("rope_kv_store_planes");
}

int semipd_rmsnorm_quant_fp8(void* out, const void* input, const void* weight, void* q, float* qs, int64_t num_tokens,
                             int64_t hidden, int64_t in_stride, int64_t out_stride, float eps, int group_size, float q_eps,
                             int dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && hidden > 0 && num_tokens < (1ll << 31) && in_stride >= hidden && out_stride >= hidden,
                   SEMIPD_EINVAL, "rmsnorm_quant_fp8: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(out && input && weight && q && qs, SEMIPD_EINVAL, "rmsnorm_quant_fp8: null pointer");
  SEMIPD_CHECK_ARG(group_size == 64 || group_size == 128 || group_size == 256 || group_size == 512, SEMIPD_ESHAPE,
                   "rmsnorm_quant_fp8: group size %d is not one of 64, 128, 256, 512", group_size);
  SEMIPD_CHECK_ARG(hidden % group_size == 0 && hidden <= 8 * 512 * 2, SEMIPD_ESHAPE,
                   "rmsnorm_quant_fp8: hidden size %lld must be a multiple of the group size and at most 8192", (long long)hidden);
  SEMIPD_CHECK_ARG(dtype == SEMIPD_BF16 || dtype == SEMIPD_F16, SEMIPD_EDTYPE, "rmsnorm_quant_fp8: bf16 / f16 only");
  SEMIPD_CHECK_ARG(aligned16(out) && aligned16(input) && aligned16(weight) && in_stride % 8 == 0 && out_stride % 8 == 0 &&
                       (reinterpret_cast<uintptr_t>(q) & 7u) == 0,
                   SEMIPD_EALIGN, "rmsnorm_quant_fp8: unaligned pointer / row stride");
  const int nvec = (int)(hidden / 8), lpg = group_size / 8;
  // the launch shape of launch_rmsnorm: same reduction order, same bits as the unfused kernel
  const int threads = rms_threads(nvec);
  const int per = (nvec + threads - 1) / threads;
  dim3 grid((unsigned)num_tokens), block(threads);
  hipStream_t st = as_stream(stream);
#define RNQ(T, MV)                                                                                                \
  hipLaunchKernelGGL((rmsnorm_vec_kernel<T, MV, false, true>), grid, block, 0, st, (T*)out, (T*)const_cast<void*>(input), \
                     (T*)nullptr, (const T*)weight, in_stride, out_stride, nvec, (int)hidden, eps, (uint8_t*)q, qs, lpg, q_eps)
  if (dtype == SEMIPD_BF16) {
    if (per <= 1) RNQ(bf16_t, 1);
    else if (per <= 2) RNQ(bf16_t, 2);
    else RNQ(bf16_t, 4);
  } else {
    if (per <= 1) RNQ(f16_t, 1);
    else if (per <= 2) RNQ(f16_t, 2);
    else RNQ(f16_t, 4);
  }
#undef RNQ
  return launch_status("rmsnorm_quant_fp8");
}

int semipd_mla_decode_prep(void* q_nope_out, void* q_input, void* kv_buf, const float* planes, int n_planes,
                           int64_t plane_elems, const int64_t* loc, const float* cos_sin_cache, const int64_t* positions,
                           const void* norm_weight, float eps, int64_t num_tokens, int num_q_heads, int nope_dim, int rope_dim,
                           int lora_rank, int64_t q_input_token_stride, int64_t q_input_head_stride, int64_t kvbuf_stride,
                           int dtype, int kv_dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && n_planes >= 1 && n_planes <= 64 && num_q_heads > 0 && nope_dim > 0 && rope_dim > 0 &&
                   lora_rank > 0, SEMIPD_EINVAL, "mla_decode_prep: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(q_nope_out && q_input && kv_buf && planes && loc && cos_sin_cache && positions && norm_weight, SEMIPD_EINVAL,
                   "mla_decode_prep: null pointer");
  SEMIPD_CHECK_ARG(dtype == SEMIPD_BF16 || dtype == SEMIPD_F16, SEMIPD_EDTYPE, "mla_decode_prep: bf16 / f16 activations");
  const int64_t row_elems = (int64_t)num_q_heads * (nope_dim + rope_dim) + lora_rank + rope_dim;
  SEMIPD_CHECK_ARG(nope_dim % 8 == 0 && rope_dim % 16 == 0 && lora_rank % 8 == 0 && lora_rank <= 512 && rope_dim <= 512 &&
                   q_input_token_stride % 8 == 0 && q_input_head_stride % 8 == 0 && kvbuf_stride % 16 == 0 &&
                   plane_elems % 4 == 0 && plane_elems >= num_tokens * row_elems && aligned16(q_nope_out) && aligned16(q_input) &&
                   aligned16(kv_buf) && aligned16(planes) && aligned16(norm_weight),
                   SEMIPD_EALIGN, "mla_decode_prep: nope %% 8, rope %% 16, lora %% 8 (<= 512), 16-byte aligned rows required");
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)num_tokens), block(256);
#define MDP(TT, KVT)                                                                                                     \
  hipLaunchKernelGGL((mla_decode_prep_kernel<TT, KVT>), grid, block, 0, st, (TT*)q_nope_out, (TT*)q_input, (KVT*)kv_buf,  \
                     planes, n_planes, plane_elems, row_elems, loc, cos_sin_cache, positions, (const TT*)norm_weight, eps, \
                     num_q_heads, nope_dim, rope_dim, lora_rank, q_input_token_stride, q_input_head_stride, kvbuf_stride)
  SEMIPD_DISPATCH_HALF(dtype, T, {
    if (kv_dtype == dtype) MDP(T, T);
    else if (kv_dtype == SEMIPD_F8E5M2) MDP(T, f8e5m2_t);
    else if (kv_dtype == SEMIPD_F8E4M3) MDP(T, f8e4m3_t);
    else {
      set_error("mla_decode_prep: unsupported kv_dtype %d", kv_dtype);
      return SEMIPD_EDTYPE;
    }
  });
#undef MDP
  return launch_status("mla_decode_prep");
}

int semipd_mla_decode_prep_rows(void* q_input, void* kv_buf, const void* q, const void* latent, const int64_t* loc,
                                const float* cos_sin_cache, const int64_t* positions, const void* norm_weight, float eps,
                                int64_t num_tokens, int num_q_heads, int nope_dim, int rope_dim, int lora_rank,
                                int64_t q_token_stride, int64_t q_head_stride, int64_t latent_token_stride,
                                int64_t q_input_token_stride, int64_t q_input_head_stride, int64_t kvbuf_stride, int dtype,
                                int kv_dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && num_q_heads > 0 && nope_dim > 0 && rope_dim > 0 && lora_rank > 0, SEMIPD_EINVAL,
                   "mla_decode_prep_rows: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(q_input && kv_buf && q && latent && loc && cos_sin_cache && positions && norm_weight, SEMIPD_EINVAL,
                   "mla_decode_prep_rows: null pointer");
  SEMIPD_CHECK_ARG(dtype == SEMIPD_BF16 || dtype == SEMIPD_F16, SEMIPD_EDTYPE, "mla_decode_prep_rows: bf16 / f16 activations");
  SEMIPD_CHECK_ARG(nope_dim % 8 == 0 && rope_dim % 16 == 0 && lora_rank % 8 == 0 && lora_rank <= 512 && rope_dim <= 512 &&
                   q_token_stride % 8 == 0 && q_head_stride % 8 == 0 && latent_token_stride % 8 == 0 &&
                   q_input_token_stride % 8 == 0 && q_input_head_stride % 8 == 0 && kvbuf_stride % 16 == 0 && aligned16(q) &&
                   aligned16(latent) && aligned16(q_input) && aligned16(kv_buf) && aligned16(norm_weight),
                   SEMIPD_EALIGN, "mla_decode_prep_rows: nope %% 8, rope %% 16, lora %% 8 (<= 512), 16-byte aligned rows required");
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)num_tokens), block(256);
#define MDR(TT, KVT)                                                                                                       \
  hipLaunchKernelGGL((mla_decode_prep_kernel<TT, KVT>), grid, block, 0, st, (TT*)nullptr, (TT*)q_input, (KVT*)kv_buf,       \
                     (const float*)nullptr, 0, (int64_t)0, (int64_t)0, loc, cos_sin_cache, positions, (const TT*)norm_weight, \
                     eps, num_q_heads, nope_dim, rope_dim, lora_rank, q_input_token_stride, q_input_head_stride, kvbuf_stride, \
                     (const TT*)q, q_token_stride, q_head_stride, (const TT*)latent, latent_token_stride)
  SEMIPD_DISPATCH_HALF(dtype, T, {
    if (kv_dtype == dtype) MDR(T, T);
    else if (kv_dtype == SEMIPD_F8E5M2) MDR(T, f8e5m2_t);
    else if (kv_dtype == SEMIPD_F8E4M3) MDR(T, f8e4m3_t);
    else {
      set_error("mla_decode_prep_rows: unsupported kv_dtype %d", kv_dtype);
      return SEMIPD_EDTYPE;
    }
  });
#undef MDR
  return launch_status("mla_decode_prep_rows");
}

int semipd_kv_store_cvt(void* buf, const void* src, const int64_t* loc, int64_t num_tokens, int64_t row_elems,
                        int64_t buf_stride, int64_t src_stride, int dtype, int kv_dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && row_elems >= 0, SEMIPD_EINVAL, "kv_store_cvt: bad sizes");
  if (num_tokens == 0 || row_elems == 0) return 0;
  SEMIPD_CHECK_ARG(buf && src && loc, SEMIPD_EINVAL, "kv_store_cvt: null pointer");
  SEMIPD_DISPATCH_DTYPE(dtype, T, return (kv_store_cvt_dispatch<T>(buf, src, loc, num_tokens, row_elems, buf_stride, src_stride, kv_dtype, as_stream(stream))));
  return 0;
}

int semipd_kv_store(void* buf, const void* src, const int64_t* loc, int64_t num_tokens,
                    int64_t row_bytes, int64_t buf_stride_bytes, int64_t src_stride_bytes,
                    void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && row_bytes >= 0, SEMIPD_EINVAL, "kv_store: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(buf && src && loc, SEMIPD_EINVAL, "kv_store: null pointer");
  return launch_row_copy<true>(buf, src, loc, num_tokens, row_bytes, buf_stride_bytes,
                               src_stride_bytes, as_stream(stream));
}

int semipd_gather_rows(void* out, const void* in, const int64_t* index, int64_t num_rows,
                       int64_t row_bytes, int64_t in_stride_bytes, void* stream) {
  SEMIPD_CHECK_ARG(num_rows >= 0 && row_bytes >= 0, SEMIPD_EINVAL, "gather_rows: bad sizes");
  if (num_rows == 0) return 0;
  SEMIPD_CHECK_ARG(out && in && index, SEMIPD_EINVAL, "gather_rows: null pointer");
  return launch_row_copy<false>(out, in, index, num_rows, row_bytes, row_bytes, in_stride_bytes,
                                as_stream(stream));
}

int semipd_build_kv_indices(const int32_t* req_to_token, int64_t req_to_token_stride,
                            const int64_t* req_pool_indices, const void* lens, int lens_is_i64,
                            const int32_t* start, int32_t* kv_indptr, int32_t* kv_indices,
                            int64_t batch, void* stream) {
  SEMIPD_CHECK_ARG(batch >= 0 && batch < 65536, SEMIPD_EINVAL, "build_kv_indices: bad batch");
  if (batch == 0) {
    if (kv_indptr) SEMIPD_HIP(hipMemsetAsync(kv_indptr, 0, sizeof(int32_t), as_stream(stream)));
    return 0;
  }
  SEMIPD_CHECK_ARG(req_to_token && req_pool_indices && lens && kv_indptr && kv_indices, SEMIPD_EINVAL,
                   "build_kv_indices: null pointer");
  if (lens_is_i64)
    hipLaunchKernelGGL((build_kv_indices_kernel<int64_t>), dim3((unsigned)batch), dim3(256), 0,
                       as_stream(stream), req_to_token, req_to_token_stride, req_pool_indices,
                       (const int64_t*)lens, start, kv_indptr, kv_indices, batch);
  else
    hipLaunchKernelGGL((build_kv_indices_kernel<int32_t>), dim3((unsigned)batch), dim3(256), 0,
                       as_stream(stream), req_to_token, req_to_token_stride, req_pool_indices,
                       (const int32_t*)lens, start, kv_indptr, kv_indices, batch);
  return launch_status("build_kv_indices");
}

int semipd_compute_positions(const int32_t* prefix_lens, const int32_t* extend_lens,
                             int64_t* positions, int32_t* extend_start_loc, int64_t batch,
                             void* stream) {
  SEMIPD_CHECK_ARG(batch >= 0 && batch < 65536, SEMIPD_EINVAL, "compute_positions: bad batch");
  if (batch == 0) return 0;
  SEMIPD_CHECK_ARG(prefix_lens && extend_lens && positions, SEMIPD_EINVAL,
                   "compute_positions: null pointer");
  hipLaunchKernelGGL(compute_positions_kernel, dim3((unsigned)batch), dim3(256), 0, as_stream(stream),
                     prefix_lens, extend_lens, positions, extend_start_loc, batch);
  return launch_status("compute_positions");
}

int semipd_argmax(const void* logits, void* out, int64_t batch, int64_t vocab,
                  int64_t logits_stride, int dtype, int out_is_i64, void* stream) {
  SEMIPD_CHECK_ARG(batch >= 0 && vocab > 0, SEMIPD_EINVAL, "argmax: bad sizes");
  if (batch == 0) return 0;
  SEMIPD_CHECK_ARG(logits && out, SEMIPD_EINVAL, "argmax: null pointer");
  const int threads = vocab >= 65536 ? 1024 : vocab >= 4096 ? 512 : 256;
  SEMIPD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((argmax_kernel<T>), dim3((unsigned)batch), dim3(threads), 0, as_stream(stream), (const T*)logits, out, vocab, logits_stride, out_is_i64));
  return launch_status("argmax");
}

int semipd_moe_sum(void* out, const void* in, int64_t num_tokens, int topk, int64_t hidden,
                   int dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && topk > 0 && hidden > 0, SEMIPD_EINVAL, "moe_sum: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(out && in, SEMIPD_EINVAL, "moe_sum: null pointer");
  SEMIPD_DISPATCH_DTYPE(dtype, T, {
    constexpr int V = Elem<T>::kVec;
    SEMIPD_CHECK_ARG(hidden % V == 0 && aligned16(out) && aligned16(in), SEMIPD_EALIGN,
                     "moe_sum: hidden must be a multiple of %d and pointers 16-byte aligned", V);
    const int64_t nv = num_tokens * (hidden / V);
    int blocks = (int)((nv + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((moe_sum_kernel<T>), dim3(blocks), dim3(256), 0, as_stream(stream), (T*)out,
                       (const T*)in, num_tokens, topk, (int)(hidden / V));
  });
  return launch_status("moe_sum");
}

/* moe_sum followed, in registers, by the element-wise tail of DeepseekV2MoE.forward: out = T(T(T(sum_j in[t,j,:]) * scale)
 * + addend[t,:]) (scale applied only when apply_scale, addend optional) -- the roundings of the three separate launches. */
int semipd_moe_sum_scale_add(void* out, const void* in, const void* addend, int64_t num_tokens, int topk, int64_t hidden,
                             float scale, int apply_scale, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && topk > 0 && hidden > 0, SEMIPD_EINVAL, "moe_sum_scale_add: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(out && in, SEMIPD_EINVAL, "moe_sum_scale_add: null pointer");
  SEMIPD_DISPATCH_DTYPE(dtype, T, {
    constexpr int V = Elem<T>::kVec;
    SEMIPD_CHECK_ARG(hidden % V == 0 && aligned16(out) && aligned16(in) && (!addend || aligned16(addend)), SEMIPD_EALIGN,
                     "moe_sum_scale_add: hidden must be a multiple of %d and pointers 16-byte aligned", V);
    const int64_t nv = num_tokens * (hidden / V);
    int blocks = (int)((nv + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((moe_sum_kernel<T>), dim3(blocks), dim3(256), 0, as_stream(stream), (T*)out, (const T*)in,
                       num_tokens, topk, (int)(hidden / V), scale, (const T*)addend, apply_scale);
  });
  return launch_status("moe_sum_scale_add");
}

/* semipd_moe_sum_scale_add whose addend is still the K-slice planes [n_planes][num_tokens][hidden] of the GEMM that produces
 * it (semipd_stream_linear_planes: the shared experts' down_proj of a decode batch): one launch less, same bits. */
int semipd_moe_sum_scale_add_planes(void* out, const void* in, const float* add_planes, int n_planes, int64_t plane_elems,
                                    int64_t num_tokens, int topk, int64_t hidden, float scale, int apply_scale, int dtype,
                                    void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && topk > 0 && hidden > 0 && n_planes >= 1 && n_planes <= 64 &&
                   plane_elems >= num_tokens * hidden, SEMIPD_EINVAL, "moe_sum_scale_add_planes: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(out && in && add_planes, SEMIPD_EINVAL, "moe_sum_scale_add_planes: null pointer");
  SEMIPD_CHECK_ARG(hidden % 8 == 0 && plane_elems % 4 == 0 && aligned16(out) && aligned16(in) && aligned16(add_planes), SEMIPD_EALIGN,
                   "moe_sum_scale_add_planes: hidden %% 8 and 16-byte aligned pointers required");
  SEMIPD_DISPATCH_HALF(dtype, T, {
    const int64_t nv = num_tokens * (hidden / 8);
    int blocks = (int)((nv + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((moe_sum_kernel<T>), dim3(blocks), dim3(256), 0, as_stream(stream), (T*)out, (const T*)in,
                       num_tokens, topk, (int)(hidden / 8), scale, (const T*)nullptr, apply_scale, add_planes, n_planes,
                       plane_elems);
  });
  return launch_status("moe_sum_scale_add_planes");
}

}  // extern "C"
