// conv_wchain3_kernel (F(4,3) x F(4,3), the dominant kernel of the headline regime) timed alone on the 256 x 256 x 128 -> 128 layer, one hipGraph of 20
// launches; build variants with -DPN_WC3_EXP=bits (1 no MFMAs, 2 no join, 4 no finish) and -DPN_WCHAIN_EXP=bits (1 no height transform, 2 no plane
// loads, 4 no weight loads):   tools/wc3q.sh "0 0" "1 0" "2 0" "4 0" "6 0" "0 6"
#include "../../partner_amd/csrc/pn_common.hip"
#include "../../partner_amd/csrc/conv_wino4.hip"
#include "../../partner_amd/csrc/conv_wchain.hip"
#include <vector>
int main(int argc, char** argv) {
  const int B = 1, H = argc > 1 ? atoi(argv[1]) : 256, W = H, C = 128;
  const size_t nv = pn_wino4_planes_floats(B, H, W, C);
  float *vin, *vout, *w, *pw, *sc, *sh;
  hipMalloc(&vin, nv * 4); hipMalloc(&vout, nv * 4); hipMalloc(&w, (size_t)C * C * 9 * 4); hipMalloc(&sc, C * 4); hipMalloc(&sh, C * 4);
  hipMalloc(&pw, pn_conv_wino44_packed_weight_floats(C, C) * 4);
  std::vector<float> h(std::max(nv, (size_t)C * C * 9));
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.3f;
  hipMemcpy(vin, h.data(), nv * 4, hipMemcpyHostToDevice);
  for (auto& v : h) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  hipMemcpy(w, h.data(), (size_t)C * C * 9 * 4, hipMemcpyHostToDevice);
  for (int i = 0; i < C; ++i) h[i] = 1.f;
  hipMemcpy(sc, h.data(), C * 4, hipMemcpyHostToDevice);
  hipMemset(sh, 0, C * 4);
  pn_pack_conv_weight_wino44_f32(w, C, C, pw, nullptr);
  pn_conv_desc d{};
  d.batch = B; d.in_h = H; d.in_w = W; d.cin = C; d.cout = C; d.kh = d.kw = 3; d.stride = 1; d.pad_h = d.pad_w = 1; d.groups = 1;
  d.in_pixel_stride = C; d.out_pixel_stride = C; d.act = PN_ACT_RELU; d.frames_in_flight = 3;
  hipStream_t st; hipStreamCreate(&st);
  for (int i = 0; i < 5; ++i) pn_conv2d_wino44_chain_f32(&d, i & 1 ? vout : vin, pw, sc, sh, i & 1 ? vin : vout, nullptr, st);
  hipStreamSynchronize(st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 20; ++i) pn_conv2d_wino44_chain_f32(&d, i & 1 ? vout : vin, pw, sc, sh, i & 1 ? vin : vout, nullptr, st);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("WC3_EXP %d WCHAIN_EXP %d: %dx%d %d->%d: %.2f us per launch\n", PN_WC3_EXP, PN_WCHAIN_EXP, H, W, C, C, ms * 1e3 / 200);
  return 0;
}
