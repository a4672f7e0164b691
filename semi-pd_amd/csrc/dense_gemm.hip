// Prefill-sized dense layers on a CU share: out[rows, n] = x[rows, k] @ W[n, k]^T through hipBLASLt with the SOLUTION
// chosen by measurement on the compute units this process owns.
//
// The library's heuristic picks persistent stream-K kernels whose grids are sized for the 256 CUs of the device; under
// an HSA_CU_MASK (the Semi-PD prefill share) those run as two rounds and a 192-CU share is no faster than a 128-CU one
// (profiles/r02_hipblaslt_under_cu_masks.txt).  Other solutions of the same library -- plain tiled kernels whose
// workgroups the hardware spreads over whatever CUs there are -- are 15-25 % faster on such a share
// (profiles/r03_library_gemm_under_masks_tunableop.txt, r03_blaslt_probe_under_masks.txt).  So the prefill instance
// times the library's solutions once at start-up, ON ITS OWN SHARE, for each weight shape of the model at a few row
// counts, and every later call takes the winner of the nearest row count.  This is the share-aware counterpart of
// UnquantizedLinearMethod.apply -> F.linear (layers/linear.py:165-172) for batches above the streaming kernel's range;
// the reference sets its shares at entrypoints/engine.py:583-634 and leaves the GEMMs to cuBLAS.
//
// Host code around a plain library GEMM; the one kernel here compares a candidate's output with the library's own
// choice (a solution that is fast and WRONG for a shape must never win: every finalist is checked before it is timed
// for the table).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <hipblaslt/hipblaslt.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "common.h"

namespace {

using semipd::set_error;

// max |a - b| and max |b| over n 16-bit elements (bf16 or f16), as float bit patterns through atomicMax (values >= 0)
template <bool BF16>
__global__ void dg_maxdiff_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, int64_t n,
                                  unsigned int* __restrict__ out) {
  float md = 0.f, mr = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float x, y;
    if (BF16) {
      x = __uint_as_float((uint32_t)a[i] << 16);
      y = __uint_as_float((uint32_t)b[i] << 16);
    } else {
      x = __half2float(__ushort_as_half(a[i]));
      y = __half2float(__ushort_as_half(b[i]));
    }
    const float d = fabsf(x - y);
    md = (d == d) ? fmaxf(md, d) : INFINITY;   // NaN anywhere = infinitely wrong
    mr = fmaxf(mr, fabsf(y));
  }
  atomicMax(out, __float_as_uint(md));
  atomicMax(out + 1, __float_as_uint(mr));
}

struct Plan {   // one (rows, n, k, dtype, ldx, ldo, bias) problem, ready to launch
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  bool have_algo = false;
  int solution_index = -1;
  bool tuned = false;
};

struct Tuned {  // winner of one tuning run
  hipblasLtMatmulAlgo_t algo;
  int solution_index;
  float us, us_default;
  int candidates, rejected;   // solutions timed; fast ones dropped because their result disagreed with the library's choice
  std::string name;           // the library's name of the winning solution (tile sizes, stream-K or not, ...)
};

struct State {
  std::mutex mu;
  hipblasLtHandle_t handle = nullptr;
  void* workspace = nullptr;
  size_t workspace_bytes = 0;
  // the library's split-K / stream-K solutions keep partial sums in the workspace while they RUN (the mutex below only
  // covers the launch): every stream that serves GEMMs gets its own (an instance has two or three: its share, the whole
  // chip, a communication stream), allocated the first time the stream is seen
  std::map<hipStream_t, void*> stream_workspace;
  // (share, dtype, n, k) -> rows -> winner.  share = the CU count the calling process declared for the stream it is on
  // (semipd_dense_gemm_set_cus): an instance that moves between a masked stream and the whole chip keeps one table per
  // CU count, and a count nothing was tuned for gets the library's own choice
  std::map<std::tuple<int, int, int64_t, int64_t>, std::map<int64_t, Tuned>> tuned;
  int share = 0;
  // (dtype, n, k) -> candidate pool (solutions that were among the fastest at some tuned row count)
  std::map<std::tuple<int, int64_t, int64_t>, std::vector<hipblasLtMatmulAlgo_t>> pool;
  std::map<std::tuple<int, int, int64_t, int64_t, int64_t, int64_t, int64_t, int>, Plan> plans;
  int device = -1;
};

State& st() {
  static State s;
  return s;
}

hipDataType hip_type(int dtype) { return dtype == SEMIPD_BF16 ? HIP_R_16BF : HIP_R_16F; }

// (called with the mutex held) nullptr + error text when a further workspace cannot be allocated
void* workspace_of(State& s, hipStream_t stream) {
  if (s.stream_workspace.empty()) {
    s.stream_workspace[stream] = s.workspace;   // the first stream takes the one semipd_dense_gemm_init allocated
    return s.workspace;
  }
  auto it = s.stream_workspace.find(stream);
  if (it != s.stream_workspace.end()) return it->second;
  void* w = nullptr;
  if (s.stream_workspace.size() >= 16 || hipMalloc(&w, s.workspace_bytes) != hipSuccess) {
    (void)hipGetLastError();
    set_error("dense_gemm: cannot allocate a %zu-byte library workspace for stream %p (%zu streams have one)",
              s.workspace_bytes, (void*)stream, s.stream_workspace.size());
    return nullptr;
  }
  s.stream_workspace[stream] = w;
  return w;
}

int ensure_init(State& s, size_t workspace_bytes) {
  if (s.handle) return 0;
  if (hipblasLtCreate(&s.handle) != HIPBLAS_STATUS_SUCCESS) {
    set_error("dense_gemm: hipblasLtCreate failed");
    s.handle = nullptr;
    return 1;
  }
  (void)hipGetDevice(&s.device);
  s.workspace_bytes = workspace_bytes ? workspace_bytes : ((size_t)64 << 20);
  if (hipMalloc(&s.workspace, s.workspace_bytes) != hipSuccess) {
    set_error("dense_gemm: cannot allocate the %zu-byte library workspace", s.workspace_bytes);
    s.workspace = nullptr;
    return 1;
  }
  return 0;
}

// out^T (column-major n x rows) = W^T (W stored column-major k x n) * x (column-major k x rows)
int make_problem(Plan& p, int dtype, int64_t rows, int64_t n, int64_t k, int64_t ldx, int64_t ldo, bool bias) {
  const hipDataType t = hip_type(dtype);
  if (hipblasLtMatrixLayoutCreate(&p.la, t, k, n, k) || hipblasLtMatrixLayoutCreate(&p.lb, t, k, rows, ldx) ||
      hipblasLtMatrixLayoutCreate(&p.lc, t, n, rows, ldo) || hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F)) {
    set_error("dense_gemm: descriptor creation failed");
    return 1;
  }
  hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
  if (bias) {
    hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi));
    hipDataType bt = t;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
  }
  return 0;
}

void destroy_problem(Plan& p) {
  if (p.la) hipblasLtMatrixLayoutDestroy(p.la);
  if (p.lb) hipblasLtMatrixLayoutDestroy(p.lb);
  if (p.lc) hipblasLtMatrixLayoutDestroy(p.lc);
  if (p.desc) hipblasLtMatmulDescDestroy(p.desc);
  p = Plan();
}

bool heuristic_algo(State& s, Plan& p, hipblasLtMatmulAlgo_t* out) {
  hipblasLtMatmulPreference_t pref;
  if (hipblasLtMatmulPreferenceCreate(&pref)) return false;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &s.workspace_bytes,
                                        sizeof(s.workspace_bytes));
  hipblasLtMatmulHeuristicResult_t r[1];
  int got = 0;
  const bool ok = hipblasLtMatmulAlgoGetHeuristic(s.handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, r, &got) ==
                      HIPBLAS_STATUS_SUCCESS && got > 0;
  hipblasLtMatmulPreferenceDestroy(pref);
  if (ok) *out = r[0].algo;
  return ok;
}

bool supported(State& s, Plan& p, hipblasLtMatmulAlgo_t& algo) {
  float alpha = 1.f, beta = 0.f;
  size_t need = 0;
  return hipblaslt_ext::matmulIsAlgoSupported(s.handle, p.desc, &alpha, p.la, p.lb, &beta, p.lc, p.lc, algo, need) ==
             HIPBLAS_STATUS_SUCCESS && need <= s.workspace_bytes;
}

// min_us: keep timing until the timed region is at least this long (0: `reps` launches and no more).  Next to another
// process's kernels a candidate has to be watched for longer than the other side's period (a decode step) or its time
// says more about what happened to run beside it than about the candidate.
float time_algo(State& s, Plan& p, hipblasLtMatmulAlgo_t& algo, const void* w, const void* x, void* o, int reps,
                hipStream_t stream, hipEvent_t e0, hipEvent_t e1, float min_us = 0.f) {
  float alpha = 1.f, beta = 0.f;
  for (int i = 0; i < 2; ++i)
    if (hipblasLtMatmul(s.handle, p.desc, &alpha, w, p.la, x, p.lb, &beta, o, p.lc, o, p.lc, &algo, s.workspace,
                        s.workspace_bytes, stream) != HIPBLAS_STATUS_SUCCESS)
      return 1e30f;
  float total_us = 0.f;
  int total_reps = 0;
  for (int round = 0; round < 3; ++round) {
    (void)hipEventRecord(e0, stream);
    for (int i = 0; i < reps; ++i)
      hipblasLtMatmul(s.handle, p.desc, &alpha, w, p.la, x, p.lb, &beta, o, p.lc, o, p.lc, &algo, s.workspace,
                      s.workspace_bytes, stream);
    (void)hipEventRecord(e1, stream);
    if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    total_us += ms * 1e3f;
    total_reps += reps;
    if (total_us >= min_us) break;
    const float per = std::max(total_us / total_reps, 1.f);
    reps = (int)std::min(4096.f, (min_us - total_us) / per + 1.f);
  }
  return total_us / total_reps;
}

float env_us(const char* name) {
  const char* v = getenv(name);
  return v ? (float)atof(v) : 0.f;
}

}  // namespace

extern "C" {

/* The CU count of the stream the following semipd_dense_gemm / _tune calls of this process run on (0 = unspecified).  Tuning
 * results are filed under it and looked up by it. */
int semipd_dense_gemm_set_cus(int cus) {
  SEMIPD_CHECK_ARG(cus >= 0 && cus <= 4096, SEMIPD_EINVAL, "dense_gemm_set_cus: bad CU count %d", cus);
  State& s = st();
  std::lock_guard<std::mutex> g(s.mu);
  s.share = cus;
  return 0;
}

int semipd_dense_gemm_init(size_t workspace_bytes) {
  State& s = st();
  std::lock_guard<std::mutex> g(s.mu);
  return ensure_init(s, workspace_bytes);
}

/* Time the library's solutions for out[rows, n] = x[rows, k] @ W[n, k]^T at each of `rows[0..num_rows)` on the CUs this
 * process owns and remember the winner per row count.  Candidates at every row count: the library's first
 * `num_heuristics` heuristic results (its ranking assumes the whole device; on a masked share the winner is typically
 * far down that list: profiles/r03_blaslt_probe_under_masks.txt) and, for the first `num_full_search` row counts, EVERY
 * solution the library supports for the problem (~2000 for bf16; 0 here turns it off, ModelRunner.tune_dense_gemms asks
 * for the first two row counts: 11-28 s of a masked prefill instance's start-up, once per (arch, CUs, library, shapes) --
 * the table is cached next to the model, semipd_dense_gemm_import) whose best
 * few then join the candidates of the remaining row counts.  Operands are scratch buffers allocated and freed here
 * (start-up only); the calling thread's current device is used. */
int semipd_dense_gemm_tune(int64_t n, int64_t k, const int64_t* rows, int num_rows, int num_full_search, int dtype,
                           int num_heuristics, int max_solutions, void* stream) {
  SEMIPD_CHECK_ARG(n > 0 && k > 0 && rows && num_rows > 0 && (dtype == SEMIPD_BF16 || dtype == SEMIPD_F16),
                   SEMIPD_EINVAL, "dense_gemm_tune: bad arguments");
  State& s = st();
  std::lock_guard<std::mutex> g(s.mu);
  if (ensure_init(s, 0)) return 1;
  hipStream_t hs = (hipStream_t)stream;
  int64_t max_rows = 0;
  for (int i = 0; i < num_rows; ++i) max_rows = std::max(max_rows, rows[i]);
  void *x = nullptr, *w = nullptr, *o = nullptr, *o_ref = nullptr;
  unsigned int* d_cmp = nullptr;
  if (hipMalloc(&x, max_rows * k * 2) || hipMalloc(&w, n * k * 2) || hipMalloc(&o, max_rows * n * 2) ||
      hipMalloc(&o_ref, max_rows * n * 2) || hipMalloc((void**)&d_cmp, 8)) {
    set_error("dense_gemm_tune: scratch allocation failed");
    if (x) (void)hipFree(x);
    if (w) (void)hipFree(w);
    if (o) (void)hipFree(o);
    if (o_ref) (void)hipFree(o_ref);
    return 1;
  }
  {  // operands with realistic bit patterns: the clock a GEMM sustains depends on its data
    std::vector<uint16_t> h((size_t)std::max(max_rows * k, n * k));
    uint32_t r = 2463534242u;
    const uint16_t base = dtype == SEMIPD_BF16 ? 0x3c00 : 0x2000;
    for (auto& v : h) {
      r ^= r << 13, r ^= r >> 17, r ^= r << 5;
      v = (uint16_t)(base | ((r >> 16) & 0xff) | ((r >> 9) & 0x8000));
    }
    (void)hipMemcpy(x, h.data(), max_rows * k * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, h.data(), n * k * 2, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const auto key = std::make_tuple(dtype, n, k);
  std::vector<hipblasLtMatmulAlgo_t>& pool = s.pool[key];
  const int nh = std::max(1, std::min(num_heuristics > 0 ? num_heuristics : 64, 256));
  int rc = 0;
  for (int ri = 0; ri < num_rows && rc == 0; ++ri) {
    const int64_t m = rows[ri];
    Plan p;
    if (make_problem(p, dtype, m, n, k, k, n, false)) { rc = 1; break; }
    std::vector<hipblasLtMatmulAlgo_t> cand;
    // SEMIPD_DG_EXCLUDE=<substring>: solutions whose library name contains it are not candidates (e.g. the persistent
    // stream-K kernels, which hold every CU they run on for the whole GEMM: next to another instance's short kernels a
    // tiled solution of equal speed is the better neighbour).  The library's own first choice is always kept.
    const char* exclude = getenv("SEMIPD_DG_EXCLUDE");
    auto add = [&](const hipblasLtMatmulAlgo_t& a) {
      hipblasLtMatmulAlgo_t c = a;
      const int idx = hipblaslt_ext::getIndexFromAlgo(c);
      if (exclude && *exclude && !cand.empty() &&
          hipblaslt_ext::getSolutionNameFromAlgo(s.handle, c).find(exclude) != std::string::npos)
        return;
      for (auto& q : cand)
        if (hipblaslt_ext::getIndexFromAlgo(q) == idx) return;
      cand.push_back(a);
    };
    {  // the library's ranking, first entry = what it would run by itself
      hipblasLtMatmulPreference_t pref;
      if (hipblasLtMatmulPreferenceCreate(&pref) == HIPBLAS_STATUS_SUCCESS) {
        hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &s.workspace_bytes,
                                              sizeof(s.workspace_bytes));
        std::vector<hipblasLtMatmulHeuristicResult_t> hr(nh);
        int got = 0;
        if (hipblasLtMatmulAlgoGetHeuristic(s.handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, nh, hr.data(), &got) ==
            HIPBLAS_STATUS_SUCCESS)
          for (int i = 0; i < got; ++i) add(hr[i].algo);
        hipblasLtMatmulPreferenceDestroy(pref);
      }
    }
    const bool have_default = !cand.empty();
    const float t_default =   // o_ref = its output
        have_default ? time_algo(s, p, cand[0], w, x, o_ref, 8, hs, e0, e1, env_us("SEMIPD_DG_FINAL_US")) : 1e30f;
    // a finalist must reproduce the library's own result on these operands up to the rounding of a different summation
    // order: max |difference| <= 2 % of max |reference| (garbage, a half-written tile or a NaN is far outside)
    auto agrees = [&](hipblasLtMatmulAlgo_t& a) -> bool {
      if (!have_default) return true;
      float alpha = 1.f, beta = 0.f;
      (void)hipMemsetAsync(o, 0xff, (size_t)m * n * 2, hs);          // NaN pattern: unwritten elements show up
      if (hipblasLtMatmul(s.handle, p.desc, &alpha, w, p.la, x, p.lb, &beta, o, p.lc, o, p.lc, &a, s.workspace,
                          s.workspace_bytes, hs) != HIPBLAS_STATUS_SUCCESS)
        return false;
      (void)hipMemsetAsync(d_cmp, 0, 8, hs);
      if (dtype == SEMIPD_BF16)
        hipLaunchKernelGGL(dg_maxdiff_kernel<true>, dim3(1024), dim3(256), 0, hs, (const uint16_t*)o, (const uint16_t*)o_ref,
                           (int64_t)m * n, d_cmp);
      else
        hipLaunchKernelGGL(dg_maxdiff_kernel<false>, dim3(1024), dim3(256), 0, hs, (const uint16_t*)o, (const uint16_t*)o_ref,
                           (int64_t)m * n, d_cmp);
      unsigned int h[2] = {0, 0};
      if (hipMemcpyAsync(h, d_cmp, 8, hipMemcpyDeviceToHost, hs) != hipSuccess || hipStreamSynchronize(hs) != hipSuccess)
        return false;
      float md, mr;
      memcpy(&md, &h[0], 4);
      memcpy(&mr, &h[1], 4);
      return md <= 0.02f * mr;
    };
    for (auto& a : pool)
      if (supported(s, p, a)) add(a);
    if (ri < num_full_search) {
      std::vector<hipblasLtMatmulHeuristicResult_t> all;
      hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
      if (hipblaslt_ext::getAllAlgos(s.handle, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, ta, tb, hip_type(dtype), hip_type(dtype),
                                     hip_type(dtype), hip_type(dtype), HIPBLAS_COMPUTE_32F, all) == HIPBLAS_STATUS_SUCCESS) {
        int tried = 0;
        for (auto& r : all) {
          if (!supported(s, p, r.algo)) continue;
          if (max_solutions > 0 && tried >= max_solutions) break;
          ++tried;
          add(r.algo);
        }
      }
    }
    // SEMIPD_DG_FIRST_US / SEMIPD_DG_FINAL_US / SEMIPD_DG_FINALISTS: minimum timed microseconds per candidate in the two
    // passes and the number of finalists -- set by the engine when the timing runs next to a busy decode instance
    const float first_us = env_us("SEMIPD_DG_FIRST_US"), final_us = env_us("SEMIPD_DG_FINAL_US");
    const size_t n_final = (size_t)std::max(6.f, std::min(32.f, env_us("SEMIPD_DG_FINALISTS")));
    std::vector<std::pair<float, int>> timed;
    for (size_t i = 0; i < cand.size(); ++i) {
      const float t = time_algo(s, p, cand[i], w, x, o, 3, hs, e0, e1, first_us);
      if (t < 1e29f) timed.push_back({t, (int)i});
    }
    std::sort(timed.begin(), timed.end());
    // the leaders again, properly -- and only those whose result agrees with the library's own choice
    float best = 1e30f;
    int best_i = -1;
    std::vector<int> finalists;
    int rejected = 0;
    for (size_t j = 0; j < timed.size() && finalists.size() < n_final; ++j) {
      if (agrees(cand[timed[j].second])) finalists.push_back(timed[j].second);
      else ++rejected;
    }
    // the library's own choice is always a finalist: a short first pass next to another process's kernels is noisy, and
    // the winner must at least be measured against it over the long pass
    if (have_default && std::find(finalists.begin(), finalists.end(), 0) == finalists.end()) finalists.push_back(0);
    for (int ci : finalists) {
      const float t = time_algo(s, p, cand[ci], w, x, o, 12, hs, e0, e1, final_us);
      if (t < best) best = t, best_i = ci;
    }
    if (ri < num_full_search)   // what a full search found joins the candidates of the other row counts
      for (int ci : finalists) {
        bool known = false;
        const int idx = hipblaslt_ext::getIndexFromAlgo(cand[ci]);
        for (auto& a : pool) known = known || hipblaslt_ext::getIndexFromAlgo(a) == idx;
        if (!known) pool.push_back(cand[ci]);
      }
    if (best_i >= 0) {
      Tuned t;
      t.algo = cand[best_i];
      t.solution_index = hipblaslt_ext::getIndexFromAlgo(t.algo);
      t.us = best;
      t.us_default = t_default;
      t.candidates = (int)timed.size();
      t.rejected = rejected;
      t.name = hipblaslt_ext::getSolutionNameFromAlgo(s.handle, t.algo);
      s.tuned[std::make_tuple(s.share, dtype, n, k)][m] = t;
    }
    destroy_problem(p);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(x);
  (void)hipFree(w);
  (void)hipFree(o);
  (void)hipFree(o_ref);
  (void)hipFree(d_cmp);
  for (auto& kv : s.plans) destroy_problem(kv.second);   // plans made before this tuning may hold other choices
  s.plans.clear();
  return rc;
}

/* out[rows, n] = x[rows, k] @ weight[n, k]^T (+ bias[n]) with the solution measured fastest on this process's CUs at the
 * nearest tuned row count (semipd_dense_gemm_tune), the library's own heuristic choice when nothing was tuned for
 * (dtype, n, k) or the winner does not support this row count.  Replaces F.linear in UnquantizedLinearMethod.apply
 * (layers/linear.py:165-172).  Returns SEMIPD_ESHAPE-free: any rows > 0. */
int semipd_dense_gemm(void* out, const void* x, const void* weight, const void* bias, int64_t rows, int64_t n, int64_t k,
                      int64_t ldx, int64_t ldo, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(rows >= 0 && n > 0 && k > 0 && ldx >= k && ldo >= n, SEMIPD_EINVAL, "dense_gemm: bad sizes");
  if (rows == 0) return 0;
  SEMIPD_CHECK_ARG(out && x && weight, SEMIPD_EINVAL, "dense_gemm: null pointer");
  SEMIPD_CHECK_ARG(dtype == SEMIPD_BF16 || dtype == SEMIPD_F16, SEMIPD_EDTYPE, "dense_gemm: bf16 / f16 only");
  State& s = st();
  std::lock_guard<std::mutex> g(s.mu);
  if (!s.handle) {
    set_error("dense_gemm: call semipd_dense_gemm_init (or _tune) first: no allocation happens on the serving path");
    return SEMIPD_EINVAL;
  }
  const auto pkey = std::make_tuple(s.share, dtype, rows, n, k, ldx, ldo, bias ? 1 : 0);
  auto it = s.plans.find(pkey);
  if (it == s.plans.end()) {
    Plan p;
    if (make_problem(p, dtype, rows, n, k, ldx, ldo, bias != nullptr)) return 1;
    auto tk = s.tuned.find(std::make_tuple(s.share, dtype, n, k));
    // a measured winner was checked against the library's own choice in ONE layout (dense rows, no bias): only calls in
    // that layout are routed to it, everything else keeps the library's choice
    if (tk != s.tuned.end() && !tk->second.empty() && ldx == k && ldo == n && !bias) {
      const Tuned* bestt = nullptr;
      double bestd = 1e30;
      for (auto& kv : tk->second) {
        const double d = std::fabs(std::log2((double)kv.first) - std::log2((double)rows));
        if (d < bestd) bestd = d, bestt = &kv.second;
      }
      hipblasLtMatmulAlgo_t a = bestt->algo;
      if (supported(s, p, a)) p.algo = a, p.have_algo = true, p.tuned = true, p.solution_index = bestt->solution_index;
    }
    if (!p.have_algo) {
      if (!heuristic_algo(s, p, &p.algo)) {
        destroy_problem(p);
        set_error("dense_gemm: the library has no solution for rows=%lld n=%lld k=%lld", (long long)rows, (long long)n,
                  (long long)k);
        return SEMIPD_ESHAPE;
      }
      p.have_algo = true;
      p.solution_index = hipblaslt_ext::getIndexFromAlgo(p.algo);
    }
    if (s.plans.size() >= 4096) {   // every distinct batch row count makes a plan: bound the cache of a long-running server
      for (auto& kv : s.plans) destroy_problem(kv.second);
      s.plans.clear();
    }
    it = s.plans.emplace(pkey, p).first;
  }
  Plan& p = it->second;
  if (bias) hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
  float alpha = 1.f, beta = 0.f;
  void* const ws = workspace_of(s, (hipStream_t)stream);
  if (!ws) return 1;
  const hipblasStatus_t rc = hipblasLtMatmul(s.handle, p.desc, &alpha, weight, p.la, x, p.lb, &beta, out, p.lc, out, p.lc,
                                             &p.algo, ws, s.workspace_bytes, (hipStream_t)stream);
  if (rc != HIPBLAS_STATUS_SUCCESS) {
    set_error("dense_gemm: hipblasLtMatmul failed with status %d (solution %d)", (int)rc, p.solution_index);
    return 1;
  }
  return 0;
}

/* Text report of the tuning table: one line per (dtype, n, k, rows): solution index, its time and the time of the library's
 * own choice on this share (microseconds).  Returns the number of bytes needed (including the terminator). */
size_t semipd_dense_gemm_report(char* buf, size_t len) {
  State& s = st();
  std::lock_guard<std::mutex> g(s.mu);
  std::string out;
  char line[1024];
  for (auto& kv : s.tuned)
    for (auto& rv : kv.second) {
      snprintf(line, sizeof(line),
               "cus=%d dtype=%d n=%lld k=%lld rows=%lld solution=%d us=%.1f library_choice_us=%.1f candidates=%d wrong_results_rejected=%d kernel=%s\n",
               std::get<0>(kv.first), std::get<1>(kv.first), (long long)std::get<2>(kv.first), (long long)std::get<3>(kv.first), (long long)rv.first,
               rv.second.solution_index, rv.second.us, rv.second.us_default, rv.second.candidates, rv.second.rejected,
               rv.second.name.c_str());
      out += line;
    }
  if (buf && len) {
    const size_t c = std::min(len - 1, out.size());
    memcpy(buf, out.data(), c);
    buf[c] = 0;
  }
  return out.size() + 1;
}

/* Load a tuning table written by semipd_dense_gemm_report (same text, one line per entry) instead of timing again: every
 * line's solution index is turned back into an algorithm of THIS library build and checked for support on the line's own
 * problem; lines that do not survive are skipped.  *loaded (may be NULL) receives the number of entries taken.  The caller
 * keys the file on (architecture, CU count, library version): an index means nothing to another build. */
int semipd_dense_gemm_import(const char* text, int* loaded) {
  SEMIPD_CHECK_ARG(text, SEMIPD_EINVAL, "dense_gemm_import: null text");
  State& s = st();
  std::lock_guard<std::mutex> g(s.mu);
  if (ensure_init(s, 0)) return 1;
  int taken = 0;
  const char* p = text;
  while (*p) {
    const char* e = strchr(p, '\n');
    std::string line(p, e ? (size_t)(e - p) : strlen(p));
    p = e ? e + 1 : p + line.size();
    int cus = 0, dtype = 0, sol = -1, cand = 0, rej = 0;
    long long n = 0, k = 0, rows = 0;
    float us = 0.f, us_def = 0.f;
    if (sscanf(line.c_str(), "cus=%d dtype=%d n=%lld k=%lld rows=%lld solution=%d us=%f library_choice_us=%f candidates=%d "
               "wrong_results_rejected=%d", &cus, &dtype, &n, &k, &rows, &sol, &us, &us_def, &cand, &rej) != 10)
      continue;
    if ((dtype != SEMIPD_BF16 && dtype != SEMIPD_F16) || n <= 0 || k <= 0 || rows <= 0 || sol < 0) continue;
    std::vector<int> idx{sol};
    std::vector<hipblasLtMatmulHeuristicResult_t> res;
    if (hipblaslt_ext::getAlgosFromIndex(s.handle, idx, res) != HIPBLAS_STATUS_SUCCESS || res.empty()) continue;
    Plan pl;
    if (make_problem(pl, dtype, rows, n, k, k, n, false)) continue;
    hipblasLtMatmulAlgo_t a = res[0].algo;
    const bool ok = supported(s, pl, a);
    destroy_problem(pl);
    if (!ok) continue;
    // an index is only meaningful inside one build of the library: the table carries the winner's kernel name, and an
    // index that names another kernel here (a table written by another hipBLASLt, a file someone else put there) is skipped
    const std::string have = hipblaslt_ext::getSolutionNameFromAlgo(s.handle, a);
    const size_t kpos = line.find(" kernel=");
    if (kpos != std::string::npos && line.substr(kpos + 8) != have) continue;
    Tuned t;
    t.algo = a;
    t.solution_index = sol;
    t.us = us;
    t.us_default = us_def;
    t.candidates = cand;
    t.rejected = rej;
    t.name = have;
    s.tuned[std::make_tuple(cus, dtype, (int64_t)n, (int64_t)k)][(int64_t)rows] = t;
    ++taken;
  }
  for (auto& kv : s.plans) destroy_problem(kv.second);
  s.plans.clear();
  if (loaded) *loaded = taken;
  return 0;
}

/* hipblasLtGetVersion of the library this process runs (part of the key of a cached tuning table: solution indices
 * are per build). */
int semipd_dense_gemm_library_version(int* version) {
  SEMIPD_CHECK_ARG(version, SEMIPD_EINVAL, "dense_gemm_library_version: null pointer");
  State& s = st();
  std::lock_guard<std::mutex> g(s.mu);
  if (ensure_init(s, 0)) return 1;
  int v = 0;
  if (hipblasLtGetVersion(s.handle, &v) != HIPBLAS_STATUS_SUCCESS) v = -1;
  *version = v;
  return 0;
}

}  // extern "C"
