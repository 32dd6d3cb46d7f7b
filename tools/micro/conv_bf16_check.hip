// conv_bf16.hip alone: results against a naive fp32 kernel on the same bf16 operands, and launch times on the layers of the Waymo PARTNER
// config (bs = 2).  Build + run on the GPU box:  bash tools/convbf16q.sh   (PN_CONV_BF16_TILE=k forces a tile)
#include "../../partner_amd/csrc/conv_bf16.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void ref_conv_kernel(const unsigned short* in, const float* w, const float* scale, const float* shift, float* out, int B, int H, int W, int OH,
                                int OW, int cin, int cout, int kh, int kw, int stride, int pad, int relu, int deconv) {
  const long long total = (long long)B * OH * OW * (deconv ? 4 * cout : cout);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int N = deconv ? 4 * cout : cout;
    const int n = (int)(i % N);
    const long long m = i / N;
    const int ox = (int)(m % OW), oy = (int)((m / OW) % OH), b = (int)(m / ((long long)OW * OH));
    float acc = 0.f;
    for (int dy = 0; dy < kh; ++dy)
      for (int dx = 0; dx < kw; ++dx) {
        const int iy = oy * stride - pad + dy, ix = ox * stride - pad + dx;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const unsigned short* px = in + ((size_t)(b * H + iy) * W + ix) * cin;
        for (int c = 0; c < cin; ++c) {
          const float x = __builtin_bit_cast(float, (unsigned)px[c] << 16);
          const float wv = __builtin_bit_cast(float, (unsigned)f32_to_bf16_bits(w[(((size_t)n * cin + c) * kh + dy) * kw + dx]) << 16);
          acc = fmaf(x, wv, acc);
        }
      }
    const int co = deconv ? n % cout : n;
    float v = acc * scale[co] + shift[co];
    if (relu) v = fmaxf(v, 0.f);
    size_t o;
    if (deconv) {
      const int quad = n / cout;
      o = (((size_t)b * 2 * OH + 2 * oy + (quad >> 1)) * (2 * OW) + 2 * ox + (quad & 1)) * cout + co;
    } else {
      o = (size_t)m * cout + n;
    }
    out[o] = v;
  }
}

struct Case { const char* name; int B, H, W, cin, cout, k, stride, deconv, f32out; };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const size_t ncases = argc > 2 ? (size_t)atoi(argv[2]) : 99;
  std::vector<Case> cases = {
      {"rpn 256x144 128->128 k3", 2, 256, 144, 128, 128, 3, 1, 0, 0},
      {"rpn 256x144 256->128 k3", 2, 256, 144, 256, 128, 3, 1, 0, 0},
      {"rpn 256x144->128x72 128->256 k3 s2", 2, 256, 144, 128, 256, 3, 2, 0, 0},
      {"rpn 128x72 256->256 k3", 2, 128, 72, 256, 256, 3, 1, 0, 0},
      {"rpn 256x144 128->256 k1", 2, 256, 144, 128, 256, 1, 1, 0, 0},
      {"rpn 128x72 256->256 deconv2x2", 2, 128, 72, 256, 256, 1, 1, 1, 0},
      {"head 256x144 512->256 k3", 2, 256, 144, 512, 256, 3, 1, 0, 0},
      {"head 256x144 256->256 k3", 2, 256, 144, 256, 256, 3, 1, 0, 0},
      {"head 256x144 512->64 k3", 2, 256, 144, 512, 64, 3, 1, 0, 0},
      {"head 256x144 256->64 k3 f32out", 2, 256, 144, 256, 64, 3, 1, 0, 1},
      {"odd 3x37x53 64->48 k3 s1", 3, 37, 53, 64, 48, 3, 1, 0, 1},
      {"odd 1x19x23 128->80 k3 s2", 1, 19, 23, 128, 80, 3, 2, 0, 0},
  };
  if (cases.size() > ncases) cases.resize(ncases);
  for (const Case& c : cases) {
    const int pad = c.k == 3 ? 1 : 0;
    const int OH = (c.H + 2 * pad - c.k) / c.stride + 1, OW = (c.W + 2 * pad - c.k) / c.stride + 1;
    const int N = c.deconv ? 4 * c.cout : c.cout;
    const size_t n_in = (size_t)c.B * c.H * c.W * c.cin, n_w = (size_t)N * c.cin * c.k * c.k;
    const size_t n_out = c.deconv ? (size_t)c.B * 4 * OH * OW * c.cout : (size_t)c.B * OH * OW * c.cout;
    std::vector<unsigned short> h_in(n_in);
    std::vector<float> h_w(n_w), h_sc(c.cout), h_sh(c.cout);
    unsigned s = 12345u + c.cin * 7 + c.cout;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto& v : h_in) v = f32_to_bf16_bits(rnd());
    const float ws = 1.f / sqrtf((float)c.cin * c.k * c.k / 3.f);
    for (auto& v : h_w) v = rnd() * ws;
    for (int i = 0; i < c.cout; ++i) { h_sc[i] = 0.75f + 0.5f * fabsf(rnd()); h_sh[i] = 0.3f * rnd(); }
    unsigned short* d_in; float *d_w, *d_sc, *d_sh, *d_ref; void *d_out, *d_pk;
    CK(hipMalloc(&d_in, n_in * 2)); CK(hipMalloc(&d_w, n_w * 4)); CK(hipMalloc(&d_sc, c.cout * 4)); CK(hipMalloc(&d_sh, c.cout * 4));
    CK(hipMalloc(&d_ref, n_out * 4)); CK(hipMalloc(&d_out, n_out * 4));
    const size_t pk = pn_conv_bf16_rows_packed_elems(N, c.cin, c.k, c.k);
    CK(hipMalloc(&d_pk, pk * 2));
    CK(hipMemcpy(d_in, h_in.data(), n_in * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, h_w.data(), n_w * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sc, h_sc.data(), c.cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_sh, h_sh.data(), c.cout * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0xff, n_out * 4));
    if (pn_pack_conv_weight_bf16_rows(d_w, N, c.cin, c.k, c.k, d_pk, nullptr) != 0) { printf("pack failed\n"); return 1; }
    pn_conv_desc d{};
    d.batch = c.B; d.in_h = c.H; d.in_w = c.W; d.cin = c.cin; d.cout = c.cout; d.groups = 1; d.kh = d.kw = c.k; d.stride = c.stride; d.pad_h = d.pad_w = pad;
    d.in_pixel_stride = c.cin; d.out_pixel_stride = c.cout; d.act = PN_ACT_RELU; d.deconv2x2 = c.deconv;
    hipLaunchKernelGGL(ref_conv_kernel, dim3(4096), dim3(256), 0, 0, d_in, d_w, d_sc, d_sh, d_ref, c.B, c.H, c.W, OH, OW, c.cin, c.cout, c.k, c.k, c.stride, pad, 1,
                       c.deconv);
    int rc = pn_conv2d_igemm_bf16(&d, d_in, d_pk, d_sc, d_sh, d_out, c.f32out, nullptr);
    if (rc != 0) { char buf[256]; pn_last_error(buf, sizeof buf); printf("%s: launch failed: %s\n", c.name, buf); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> h_ref(n_out), h_out(n_out);
    CK(hipMemcpy(h_ref.data(), d_ref, n_out * 4, hipMemcpyDeviceToHost));
    if (c.f32out) {
      CK(hipMemcpy(h_out.data(), d_out, n_out * 4, hipMemcpyDeviceToHost));
    } else {
      std::vector<unsigned short> t(n_out);
      CK(hipMemcpy(t.data(), d_out, n_out * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n_out; ++i) { unsigned u = (unsigned)t[i] << 16; memcpy(&h_out[i], &u, 4); }
    }
    double maxref = 0, maxerr = 0; size_t bad = 0;
    for (size_t i = 0; i < n_out; ++i) maxref = fmax(maxref, fabs(h_ref[i]));
    const double tol = c.f32out ? 2e-5 : 4.5e-3;       // f32 out: summation order only; bf16 out: half an ulp of bf16 (2^-9) + that
    for (size_t i = 0; i < n_out; ++i) {
      const double e = fabs((double)h_out[i] - h_ref[i]);
      const double lim = c.f32out ? tol * maxref : tol * fmax(fabs(h_ref[i]), 1e-2 * maxref);
      if (!(e <= lim)) ++bad;
      maxerr = fmax(maxerr, e);
    }
    // timed as the product runs it: the launches captured into ONE hipGraph (no host launch cost between the kernels)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int i = 0; i < 3; ++i) pn_conv2d_igemm_bf16(&d, d_in, d_pk, d_sc, d_sh, d_out, c.f32out, st);
    CK(hipStreamSynchronize(st));
    hipGraph_t graph; hipGraphExec_t gexec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < reps; ++i) pn_conv2d_igemm_bf16(&d, d_in, d_pk, d_sc, d_sh, d_out, c.f32out, st);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(gexec, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(gexec, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    CK(hipGraphExecDestroy(gexec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(st));
    const double flop = 2.0 * c.B * OH * OW * (double)N * c.cin * c.k * c.k;
    printf("%-40s %s  max|err| %.3g (max|ref| %.3g, %zu bad)  %8.2f us  %7.1f TFLOP/s  %.3f of 2.5 PF\n", c.name, bad ? "FAIL" : "ok  ", maxerr, maxref, bad, us,
           flop / us * 1e-6, flop / us * 1e-6 / 2500.0);
    hipFree(d_in); hipFree(d_w); hipFree(d_sc); hipFree(d_sh); hipFree(d_ref); hipFree(d_out); hipFree(d_pk);
  }
  // the token GEMMs of the SetBlock / Swin stages on the same kernel (pn_linear_bf16): times only (tests/test_hip_linear.py checks the values)
  struct Lin { int m, k, n, f32out, act, res; };
  const Lin lins[] = {{73728, 256, 256, 1, 0, 1}, {73728, 256, 512, 1, 0, 0}, {73728, 256, 768, 1, 0, 0}, {73728, 256, 1024, 0, PN_ACT_GELU, 0},
                      {73728, 1024, 256, 1, 0, 1}, {73728, 256, 256, 0, 0, 0}};
  for (const Lin& l : lins) {
    void *x, *w, *out; float *wf, *bias, *res;
    CK(hipMalloc(&x, (size_t)l.m * l.k * 2)); CK(hipMalloc(&wf, (size_t)l.n * l.k * 4)); CK(hipMalloc(&bias, l.n * 4));
    CK(hipMalloc(&res, (size_t)l.m * l.n * 4)); CK(hipMalloc(&out, (size_t)l.m * l.n * 4));
    CK(hipMemset(x, 0x3c, (size_t)l.m * l.k * 2)); CK(hipMemset(wf, 0, (size_t)l.n * l.k * 4)); CK(hipMemset(bias, 0, l.n * 4)); CK(hipMemset(res, 0, (size_t)l.m * l.n * 4));
    const size_t pk = pn_conv_bf16_rows_packed_elems(l.n, l.k, 1, 1);
    CK(hipMalloc(&w, pk * 2));
    if (pn_pack_conv_weight_bf16_rows(wf, l.n, l.k, 1, 1, w, nullptr) != 0) { printf("pack failed\n"); return 1; }
    hipStream_t st; CK(hipStreamCreate(&st));
    auto run = [&]() { return pn_linear_bf16(x, l.m, l.k, l.k, w, l.n, bias, l.act, l.res ? res : nullptr, l.n, out, l.n, l.f32out, st); };
    for (int i = 0; i < 3; ++i) if (run() != 0) { char buf[256]; pn_last_error(buf, sizeof buf); printf("linear failed: %s\n", buf); return 1; }
    CK(hipStreamSynchronize(st));
    hipGraph_t graph; hipGraphExec_t gexec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < reps; ++i) run();
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(gexec, st)); CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(gexec, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, flop = 2.0 * l.m * (double)l.k * l.n;
    const double bytes = (double)l.m * l.k * 2 + (double)l.m * l.n * (l.f32out ? 4 : 2) + (l.res ? (double)l.m * l.n * 4 : 0);
    printf("linear %d x %4d -> %4d %s%s%s  %8.2f us  %7.1f TFLOP/s  %.3f of 2.5 PF   %.2f TB/s of compulsory traffic\n", l.m, l.k, l.n, l.f32out ? "f32 out" : "bf16 out",
           l.act ? " gelu" : "", l.res ? " +res" : "", us, flop / us * 1e-6, flop / us * 1e-6 / 2500.0, bytes / us * 1e-6);
    CK(hipGraphExecDestroy(gexec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(st));
    hipFree(x); hipFree(wf); hipFree(bias); hipFree(res); hipFree(out); hipFree(w);
  }
  return 0;
}
