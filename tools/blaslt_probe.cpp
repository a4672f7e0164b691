// Which hipBLASLt solution is fastest for out[M, N] = x[M, K] @ W[N, K]^T (bf16) on the CUs THIS process owns
// (HSA_CU_MASK)?  Times the library's first heuristic choice, the best of its top-`heur` heuristic results and the best of
// every solution it has, and prints the winner's index and kernel name.
//   hipcc -O2 --offload-arch=gfx950 tools/blaslt_probe.cpp -o /tmp/blaslt_probe -lhipblaslt
//   HSA_CU_MASK=0:0-191 /tmp/blaslt_probe 1024 28672 4096 [1024 4096 14336 ...]
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <hipblaslt/hipblaslt.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                        \
  do {                                                               \
    auto _e = (x);                                                   \
    if (_e != 0) {                                                   \
      fprintf(stderr, "%s failed: %d (line %d)\n", #x, (int)_e, __LINE__); \
      exit(1);                                                       \
    }                                                                \
  } while (0)

int main(int argc, char** argv) {
  hipblasLtHandle_t h;
  CK(hipblasLtCreate(&h));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t ws_bytes = 256u << 20;
  void* ws;
  CK(hipMalloc(&ws, ws_bytes));
  const int iters = getenv("PROBE_ITERS") ? atoi(getenv("PROBE_ITERS")) : 10;
  const int heur_n = getenv("PROBE_HEUR") ? atoi(getenv("PROBE_HEUR")) : 64;
  printf("# blaslt_probe HSA_CU_MASK=%s\n", getenv("HSA_CU_MASK") ? getenv("HSA_CU_MASK") : "-");
  for (int a = 1; a + 2 < argc; a += 3) {
    const int64_t M = atoll(argv[a]), N = atoll(argv[a + 1]), K = atoll(argv[a + 2]);
    void *x, *w, *o;
    CK(hipMalloc(&x, M * K * 2));
    CK(hipMalloc(&w, N * K * 2));
    CK(hipMalloc(&o, M * N * 2));
    {  // bf16 data that is not all zero (clocks depend on the operands): 0x3c00..0x3cff ~ 0.0078..0.03
      std::vector<uint16_t> hx(M * K), hw(N * K);
      uint32_t s = 12345;
      for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 | ((s >> 16) & 0xff) | ((s >> 9) & 0x8000)); }
      for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 | ((s >> 16) & 0xff) | ((s >> 9) & 0x8000)); }
      CK(hipMemcpy(x, hx.data(), M * K * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(w, hw.data(), N * K * 2, hipMemcpyHostToDevice));
    }
    hipblasLtMatrixLayout_t la, lb, lc;
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K, N, K));   // W as a column-major K x N matrix, used transposed
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, K, M, K));   // x as column-major K x M
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, N, M, N));   // out^T column-major = out row-major
    hipblasLtMatmulDesc_t desc;
    CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    hipblasLtMatmulPreference_t pref;
    CK(hipblasLtMatmulPreferenceCreate(&pref));
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
    float alpha = 1.f, beta = 0.f;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_algo = [&](hipblasLtMatmulAlgo_t& algo, int reps) -> float {
      for (int i = 0; i < 2; ++i)
        if (hipblasLtMatmul(h, desc, &alpha, w, la, x, lb, &beta, o, lc, o, lc, &algo, ws, ws_bytes, st) != 0) return 1e30f;
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) hipblasLtMatmul(h, desc, &alpha, w, la, x, lb, &beta, o, lc, o, lc, &algo, ws, ws_bytes, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      return ms * 1e3f / reps;
    };
    std::vector<hipblasLtMatmulHeuristicResult_t> heur(heur_n);
    int got = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, heur_n, heur.data(), &got));
    float t_first = -1, t_heur = 1e30f;
    int best_heur = -1;
    for (int i = 0; i < got; ++i) {
      float t = time_algo(heur[i].algo, iters);
      if (i == 0) t_first = t;
      if (t < t_heur) t_heur = t, best_heur = i;
    }
    std::vector<hipblasLtMatmulHeuristicResult_t> all;
    CK(hipblaslt_ext::getAllAlgos(h, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, ta, tb, HIP_R_16BF, HIP_R_16BF, HIP_R_16BF,
                                  HIP_R_16BF, HIPBLAS_COMPUTE_32F, all));
    std::vector<std::pair<float, int>> timed;
    int supported = 0;
    for (size_t i = 0; i < all.size(); ++i) {
      size_t need = 0;
      if (hipblaslt_ext::matmulIsAlgoSupported(h, desc, &alpha, la, lb, &beta, lc, lc, all[i].algo, need) != 0 || need > ws_bytes)
        continue;
      ++supported;
      float t = time_algo(all[i].algo, 3);
      timed.push_back({t, (int)i});
    }
    std::sort(timed.begin(), timed.end());
    float t_all = 1e30f;
    int best_all = -1;
    for (size_t j = 0; j < std::min<size_t>(timed.size(), 8); ++j) {   // re-time the leaders properly
      float t = time_algo(all[timed[j].second].algo, iters * 2);
      if (t < t_all) t_all = t, best_all = timed[j].second;
    }
    const double fl = 2.0 * M * N * K;
    printf("M=%5lld N=%6lld K=%6lld: first %7.1f us %6.0f TF | best of %d heuristics [#%d idx %d] %7.1f us %6.0f TF | "
           "best of %d/%zu algos [idx %d] %7.1f us %6.0f TF  %s\n",
           (long long)M, (long long)N, (long long)K, t_first, fl / t_first / 1e6, got, best_heur,
           best_heur >= 0 ? hipblaslt_ext::getIndexFromAlgo(heur[best_heur].algo) : -1, t_heur, fl / t_heur / 1e6, supported,
           all.size(), best_all >= 0 ? hipblaslt_ext::getIndexFromAlgo(all[best_all].algo) : -1, t_all, fl / t_all / 1e6,
           best_all >= 0 ? hipblaslt_ext::getKernelNameFromAlgo(h, all[best_all].algo).c_str() : "");
    fflush(stdout);
    hipFree(x), hipFree(w), hipFree(o);
  }
  return 0;
}
