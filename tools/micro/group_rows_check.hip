// sparse_group_rows_kernel on a synthetic neighbour table (n sites x 27 taps, ~45 % present), phases ablated by -DPN_SG_EXP bits
// (1: no mask build, 2: no sort):  for e in 0 1 2 3; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -DPN_SG_EXP=$e -I../../include group_rows_check.hip -o /tmp/grc && /tmp/grc; done
#include "../../partner_amd/csrc/pn_common.hip"
#include "../../partner_amd/csrc/sparse_group.hip"
#include <vector>
int main() {
  const int n = 100000, cap = 150000, taps = 27;
  std::vector<int32_t> h((size_t)cap * taps);
  for (auto& v : h) v = (rand() % 100) < 45 ? rand() % n : -1;
  int32_t *nbr, *nv, *perm; uint32_t* gm;
  hipMalloc(&nbr, h.size() * 4); hipMalloc(&nv, 4); hipMalloc(&perm, cap * 4); hipMalloc(&gm, (cap / 32 + 1) * 4);
  hipMemcpy(nbr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(nv, &n, 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) pn_sparse_group_rows(nbr, nv, cap, taps, perm, gm, nullptr);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("EXP %d: %.2f us per launch\n", PN_SG_EXP, ms * 1e3 / 20);
  }
  std::vector<int32_t> p(cap);
  hipMemcpy(p.data(), perm, cap * 4, hipMemcpyDeviceToHost);
  std::vector<char> seen(cap, 0); int bad = 0;
  for (int i = 0; i < cap; ++i) { if (p[i] >= 0) { if (p[i] >= n || seen[p[i]] || p[i] / 4096 != i / 4096) ++bad; else seen[p[i]] = 1; } }
  int cnt = 0; for (int i = 0; i < n; ++i) cnt += seen[i];
  printf("perm: %d of %d sites placed, %d bad\n", cnt, n, bad);
  return 0;
}
