// Where a wave of conv_stem_kernel spends its cycles (s_memtime stamps at slab / tile boundaries only, nothing inside the k-steps):
// k-steps of a slab, wait at the slab barrier, K-half exchange, epilogue.  Builds the library's own source with -DST_DIAG.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DST_DIAG -I include -I danet-densepose2smpl_amd/csrc tools/experiments/stem_diag.hip -o tools/experiments/stem_diag
#include "../../danet-densepose2smpl_amd/csrc/capi.hip"
#include "../../danet-densepose2smpl_amd/csrc/conv_stem.hip"
#include <algorithm>
#include <vector>

int main()
{
    const int B = 768, H = 64, W = 64, C = 64, OH = 32, OW = 32;
    const size_t xb = (size_t)B * H * W * C * 2, yb = (size_t)B * OH * OW * C * 2, wb = (size_t)4 * 100 * 1024;
    void *x, *y, *w;
    hipMalloc(&x, xb); hipMalloc(&y, yb); hipMalloc(&w, wb);
    hipMemset(x, 0x3c, xb); hipMemset(w, 0x38, wb);            // finite bf16 patterns
    for (int it = 0; it < 3; ++it) {
        const int rc = danet_conv_stem_forward(x, w, y, B, H, W, C, OH, OW, C, nullptr, nullptr);
        if (rc) { printf("launch failed: %s\n", danet_last_error()); return 1; }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < 10; ++it) danet_conv_stem_forward(x, w, y, B, H, W, C, OH, OW, C, nullptr, nullptr);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> d(256 * 4 * 8);
    hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(g_stem_diag), d.size() * 8);
    const char* names[5] = {"k-steps", "slab barrier wait", "exchange", "epilogue", "life"};
    printf("{\"us_per_launch\": %.1f", ms * 100.f);
    for (int wv = 0; wv < 4; ++wv) {
        printf(", \"wave%d (pw %d, kw %d)\": {", wv, wv & 1, wv >> 1);
        for (int q = 0; q < 5; ++q) {
            std::vector<unsigned long long> v;
            for (int b = 0; b < 256; ++b) v.push_back(d[(b * 4 + wv) * 8 + q]);
            std::sort(v.begin(), v.end());
            printf("%s\"%s\": %llu", q ? ", " : "", names[q], v[128]);
        }
        printf("}");
    }
    printf(", \"note\": \"median over 256 workgroups of s_memtime ticks (100 MHz: x24 for shader cycles at 2.4 GHz), 12 tiles x 4 slabs per workgroup\"}\n");
    return 0;
}
