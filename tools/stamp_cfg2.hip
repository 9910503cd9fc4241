// Diagnostic build (never shipped): the cfg2 kernels with in-kernel wall-clock stamps (s_memrealtime, 100 MHz) per
// workgroup and phase, to see where the fixed cost of a 8-15 us kernel sits.  Builds eps_mfma.hip with -DDCTN_STAMPS
// into this executable; the library itself contains no stamp.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize \
//         -DDCTN_STAMPS tools/stamp_cfg2.hip -o tools/stamp_cfg2 && tools/stamp_cfg2 [B]
#include "../dctn_amd/csrc/eps_mfma.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

void dctn_set_last_kernel(const char*) {}

// the part of eps_fill_params (eps_generic.hip) this tool needs
int eps_fill_params(EpsP& p, const int64_t xs[5], int C, int B, int H, int W, int Q, int K, int O, int policy) {
  p.opts = policy & ~DCTN_PREC_MASK;
  p.C = C; p.B = B; p.H = H; p.W = W; p.Q = Q; p.K = K; p.O = O;
  p.N = K * K * C; p.Ho = H - K + 1; p.Wo = W - K + 1;
  p.Wn = (long long)B * p.Ho * p.Wo;
  p.R = 1;
  for (int n = 0; n < p.N; ++n) p.R *= Q;
  for (int i = 0; i < 5; ++i) p.s[i] = xs[i];
  p.m = 0; p.LO = 1; p.NH = p.N; p.HI = p.R; p.bits = 4;
  return DCTN_OK;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static void report(const char* name, const std::vector<unsigned long long>& st, int nblk, int nslots, const char* const* labels) {
  unsigned long long t0 = ~0ull;
  for (int b = 0; b < nblk; ++b) if (st[b * 8]) t0 = std::min(t0, st[b * 8]);
  printf("%s: %d workgroups; times in us relative to the first workgroup's entry (min / median / max over workgroups)\n", name, nblk);
  for (int s = 0; s < nslots; ++s) {
    std::vector<double> v;
    for (int b = 0; b < nblk; ++b) if (st[b * 8 + s]) v.push_back((double)(st[b * 8 + s] - t0) * 0.01);
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    printf("  %-34s %7.2f %7.2f %7.2f   (%zu stamps)\n", labels[s], v.front(), v[v.size() / 2], v.back(), v.size());
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1024, H = 28, W = 28, Q = 2, K = 3, O = 4, C = 1, Cout = 10;
  const int Ho = H - K + 1, P = Ho * Ho;
  const size_t nx = (size_t)B * H * W * Q, nf = (size_t)B * P * O, nc = 512 * O, nw = (size_t)Cout * P * O;
  std::vector<unsigned short> hx(nx), hc(nc), hw(nw), hdl((size_t)B * Cout);
  srand(1);
  auto bf = [](float f) { union { float f; unsigned u; } c; c.f = f; return (unsigned short)(c.u >> 16); };
  for (auto& v : hx) v = bf((float)(rand() % 4096) / 4096.f);
  for (auto& v : hc) v = bf(((float)(rand() % 4096) / 4096.f - 0.5f) * 0.1f);
  for (auto& v : hw) v = bf(((float)(rand() % 4096) / 4096.f - 0.5f) * 0.1f);
  for (auto& v : hdl) v = bf(((float)(rand() % 4096) / 4096.f - 0.5f) * 0.1f);
  void *x, *core, *feat, *wgt, *dl, *dcore, *dw, *db, *ws, *logit;
  unsigned long long* stamps;
  CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&core, nc * 2)); CK(hipMalloc(&feat, nf * 2)); CK(hipMalloc(&wgt, nw * 2));
  CK(hipMalloc(&dl, hdl.size() * 2)); CK(hipMalloc(&dcore, nc * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&db, 64)); CK(hipMemset(db, 0, 64)); CK(hipMalloc(&logit, (size_t)B * Cout * 2));
  CK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(core, hc.data(), nc * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(wgt, hw.data(), nw * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dl, hdl.data(), hdl.size() * 2, hipMemcpyHostToDevice));
  const int NB = 2048;
  CK(hipMalloc(&stamps, NB * 8 * 8));
  const int64_t xs[5] = {(int64_t)B * H * W * Q, (int64_t)H * W * Q, (int64_t)W * Q, Q, 1};
  EpsP p;
  eps_fill_params(p, xs, C, B, H, W, Q, K, O, 0);
  const size_t wsb = eps_head_bwd_mfma_workspace(p, Cout, DCTN_BF16, 0);
  CK(hipMalloc(&ws, wsb));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::vector<unsigned long long> hs(NB * 8);
  const char* fl[] = {"entry", "core staged, fragments in regs", "first sample done", "loop done", "", "", "", ""};
  const char* hl[] = {"entry", "core staged, fragments in regs", "first sample done", "last group's samples done (wave 0)", "end", "", "", ""};
  const char* bl[] = {"entry", "head-weight slice arrived", "first sample's loads arrived", "loop done", "dCore tile reduced + stored",
                      "end", "loop done, last wave", "every wave out of the loop (barrier)"};
  const char* rl[] = {"entry (dCore roles: blocks 0-63; gemm roles after)", "gemm: products done, tile in LDS", "gemm: barrier passed",
                      "", "end", "", "", ""};
  for (int rep = 0; rep < 3; ++rep) {   // the last repetition is reported (warm caches, as inside a replayed step)
    dctn_stamps_set(nullptr);
    for (int i = 0; i < 3; ++i) {
      if (eps_fwd_mfma(x, core, feat, p, DCTN_BF16, 0, st) != DCTN_OK) { printf("fwd failed\n"); return 1; }
      if (eps_head_bwd_mfma(x, feat, dl, wgt, dcore, dw, db, ws, wsb, p, Cout, DCTN_BF16, 0, st) != DCTN_OK) { printf("bwd failed\n"); return 1; }
    }
    CK(hipStreamSynchronize(st));
    CK(hipMemset(stamps, 0, NB * 8 * 8));
    dctn_stamps_set(stamps);
    if (eps_fwd_mfma(x, core, feat, p, DCTN_BF16, 0, st) != DCTN_OK) return 1;
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hs.data(), stamps, NB * 8 * 8, hipMemcpyDeviceToHost));
    if (rep == 2) report("eps_fwd_q2reg_k", hs, NB, 4, fl);
    CK(hipMemset(stamps, 0, NB * 8 * 8));
    if (eps_head_fwd_mfma(x, core, wgt, db, feat, logit, p, Cout, DCTN_BF16, 0, st) != DCTN_OK) { printf("fused fwd failed\n"); return 1; }
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hs.data(), stamps, NB * 8 * 8, hipMemcpyDeviceToHost));
    if (rep == 2) report("eps_fwd_head_q2reg_k (layer + head)", hs, NB, 5, hl);
    CK(hipMemset(stamps, 0, NB * 8 * 8));
    p.opts = DCTN_OPT_MAIN_KERNEL_ONLY;
    if (eps_head_bwd_mfma(x, feat, dl, wgt, dcore, dw, db, ws, wsb, p, Cout, DCTN_BF16, 0, st) != DCTN_PARTIAL) return 1;
    p.opts = 0;
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hs.data(), stamps, NB * 8 * 8, hipMemcpyDeviceToHost));
    if (rep == 2) report("eps_bwd_dcore_q2reg_k (fused head)", hs, NB, 8, bl);
    // the finishing kernel (eager: its own launch, after the dCore kernel has drained)
    dctn_stamps_set(nullptr);
    CK(hipMemset(stamps, 0, NB * 8 * 8));
    dctn_reduce_stamps_set(stamps);
    if (eps_head_bwd_mfma(x, feat, dl, wgt, dcore, dw, db, ws, wsb, p, Cout, DCTN_BF16, 0, st) != DCTN_OK) return 1;
    CK(hipStreamSynchronize(st));
    dctn_reduce_stamps_set(nullptr);
    CK(hipMemcpy(hs.data(), stamps, NB * 8 * 8, hipMemcpyDeviceToHost));
    if (rep == 2) {
      std::vector<unsigned long long> a(hs.begin(), hs.begin() + 64 * 8), b(hs.begin() + 64 * 8, hs.end());
      // common origin: the earliest entry of either role
      report("eps_head_reduce_k, all roles", hs, NB, 5, rl);
      report("eps_head_reduce_k, dCore roles only", a, 64, 5, rl);
    }
  }
  return 0;
}
