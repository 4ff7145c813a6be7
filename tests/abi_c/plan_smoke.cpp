// The plan layer of the C ABI used without Python or torch: plain HIP runtime + libfdm_hip.so.
// Loads a model (reference state-dict names -> fp32 arrays) and a sampling case from a flat record file, then runs
//   fdm_plan_create -> fdm_plan_set_weights -> fdm_audio_prepare -> fdm_sample_graph (DDPM with injected noise, every step
//   recorded; DDIM; Philox DDPM twice) -> fdm_denoise_step
// and compares with the expected latents in the file (the reference's own outputs, tests/golden/chains_*.npz) at 1e-4.
// Built and run by tests/test_abi_c_gpu.py:  hipcc plan_smoke.cpp -I include -L <dir> -lfdm_hip;  ./plan_smoke case.bin [dtype]
// Record format: u32 name_len, name, u64 n, n floats.  "w:<name>" = weight, anything else = case tensor.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "fdm_hip.h"

#define CK(x) do { if ((x) != 0) { printf("FAIL %s: %s\n", #x, fdm_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float* to_dev(const std::vector<float>& v) {
  float* p = nullptr;
  if (hipMalloc(&p, v.size() * 4) != hipSuccess) return nullptr;
  if (hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return p;
}
static double max_abs_diff(const float* dev, const std::vector<float>& ref, size_t off, size_t n) {
  std::vector<float> h(n);
  if (hipMemcpy(h.data(), dev, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1e30;
  double m = 0;
  for (size_t i = 0; i < n; ++i) {
    const double d = std::fabs((double)h[i] - ref[off + i]);
    if (!(d <= m)) m = d;        // also catches NaN
  }
  return m;
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: plan_smoke case.bin [dtype code]\n"); return 2; }
  if (!fdm_device_ok()) { printf("no gfx950 device\n"); return 2; }
  const int dtype = argc > 2 ? atoi(argv[2]) : FDM_F32;
  std::map<std::string, std::vector<float>> rec;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
  for (;;) {
    unsigned nl = 0;
    if (fread(&nl, 4, 1, f) != 1) break;
    std::string name(nl, '\0');
    unsigned long long n = 0;
    if (fread(&name[0], 1, nl, f) != nl || fread(&n, 8, 1, f) != 1) { printf("truncated record\n"); return 2; }
    std::vector<float> v(n);
    if (fread(v.data(), 4, n, f) != n) { printf("truncated record %s\n", name.c_str()); return 2; }
    rec[name] = std::move(v);
  }
  fclose(f);
  const std::vector<float>& meta = rec["meta"];      // [L, N, fw, n_steps]
  const int L = (int)meta[0], N = (int)meta[1], fw = (int)meta[2], T = (int)meta[3];
  char preset[64] = {0};
  for (size_t i = 0; i < rec["preset"].size() && i < 63; ++i) preset[i] = (char)rec["preset"][i];

  hipStream_t st;
  HK(hipStreamCreate(&st));
  fdm_model_desc desc;
  CK(fdm_model_preset(preset, &desc));
  fdm_plan* plan = nullptr;
  CK(fdm_plan_create(&desc, 1, L, 0, dtype, &plan));
  int nw = 0;
  for (auto& kv : rec)
    if (kv.first.rfind("w:", 0) == 0) { CK(fdm_plan_set_weights(plan, kv.first.c_str() + 2, kv.second.data(), (long long)kv.second.size(), st)); ++nw; }
  HK(hipStreamSynchronize(st));
  float* hub = to_dev(rec["hub"]); float* style = to_dev(rec["style"]); float* xT = to_dev(rec["x_T"]); float* noise = to_dev(rec["noise"]);
  const size_t n = (size_t)L * desc.d;
  float *out, *out2, *record;
  HK(hipMalloc(&out, n * 4)); HK(hipMalloc(&out2, n * 4)); HK(hipMalloc(&record, n * 4 * T));
  CK(fdm_audio_prepare(plan, hub, 1, N, fw, style, nullptr, L, 0, st));

  // DDPM chain with injected noise, every step recorded, eager and as the captured graph
  std::vector<int> ts(T);
  for (int i = 0; i < T; ++i) ts[i] = (int)rec["t_list"][i];
  fdm_sample_args a;
  memset(&a, 0, sizeof(a));
  a.kind = 0; a.x_T = xT; a.out = out; a.t_list = ts.data(); a.n_steps = T; a.noise = noise; a.record = record;
  CK(fdm_sample_graph(plan, &a, st));
  HK(hipStreamSynchronize(st));
  const double e_steps = max_abs_diff(record, rec["expected_steps"], 0, n * T);
  a.record = nullptr; a.out = out2;
  CK(fdm_sample_graph(plan, &a, st));            // multi-step graph launches
  long long launches = 0, per_step = 0;
  CK(fdm_plan_get(plan, "graph_launches", &launches));
  CK(fdm_plan_get(plan, "launches_per_step", &per_step));
  a.eager = 1; a.out = out;
  CK(fdm_sample_graph(plan, &a, st));
  HK(hipStreamSynchronize(st));
  const double e_final = max_abs_diff(out2, rec["expected_steps"], n * (T - 1), n);
  std::vector<float> h1(n), h2(n);
  HK(hipMemcpy(h1.data(), out, n * 4, hipMemcpyDeviceToHost)); HK(hipMemcpy(h2.data(), out2, n * 4, hipMemcpyDeviceToHost));
  if (memcmp(h1.data(), h2.data(), n * 4) != 0) { printf("FAIL: graph replay differs from eager launches\n"); return 1; }

  // DDIM, 3 steps (two live denoiser calls)
  memset(&a, 0, sizeof(a));
  a.kind = 1; a.x_T = xT; a.out = out; a.ddim_steps = 3;
  CK(fdm_sample_graph(plan, &a, st));
  HK(hipStreamSynchronize(st));
  const double e_ddim = max_abs_diff(out, rec["expected_ddim3"], 0, n);

  // Philox noise: two runs with one seed agree bit for bit, another seed differs
  memset(&a, 0, sizeof(a));
  a.kind = 0; a.x_T = xT; a.t_list = ts.data(); a.n_steps = T; a.seed = 42;
  a.out = out; CK(fdm_sample_graph(plan, &a, st));
  a.out = out2; CK(fdm_sample_graph(plan, &a, st));
  HK(hipStreamSynchronize(st));
  HK(hipMemcpy(h1.data(), out, n * 4, hipMemcpyDeviceToHost)); HK(hipMemcpy(h2.data(), out2, n * 4, hipMemcpyDeviceToHost));
  if (memcmp(h1.data(), h2.data(), n * 4) != 0) { printf("FAIL: Philox run not deterministic\n"); return 1; }
  a.seed = 43; CK(fdm_sample_graph(plan, &a, st));
  HK(hipStreamSynchronize(st));
  HK(hipMemcpy(h2.data(), out2, n * 4, hipMemcpyDeviceToHost));
  if (memcmp(h1.data(), h2.data(), n * 4) == 0) { printf("FAIL: seed ignored\n"); return 1; }

  CK(fdm_denoise_step(plan, xT, ts[0], 0.f, out, nullptr, st));
  HK(hipStreamSynchronize(st));
  // error paths: bad timestep, sampling before prepare on a fresh plan
  if (fdm_denoise_step(plan, xT, 1000, 0.f, out, nullptr, st) == 0) { printf("FAIL: t = 1000 accepted\n"); return 1; }
  CK(fdm_plan_destroy(plan));

  const double tol = (dtype == FDM_BF16) ? 0.15 : 1e-4;
  printf("plan_smoke: %d weights, L=%d, %d DDPM steps: max|steps - reference| = %.3e, final (graph) %.3e, DDIM-3 %.3e; %lld kernel launches per step, %lld graph launches for %d steps\n",
         nw, L, T, e_steps, e_final, e_ddim, per_step, launches, T);
  if (!(e_steps < tol && e_final < tol && e_ddim < tol)) { printf("FAIL: above tolerance %.1e\n", tol); return 1; }
  printf("plan_smoke ok\n");
  return 0;
}
