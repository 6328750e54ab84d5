// pack_tile_check.cpp -- host check of pack_tile.h (test infrastructure; built and run by tests/test_pack_tile_host.py, never linked into the library).
// Reads back-to-back dl_pack_desc records from the file named on the command line; for each one it fills a random master weight, runs the tiled form
// (pack_tile_load / pack_tile_store, thread by thread, a plain array standing in for LDS) and compares every bit of both images with the element-wise
// decode (the loop of pack_weights_kernel, restated below).  Prints one line per descriptor; exit code 0 = every eligible descriptor identical.
#include "pack_tile.h"
#include <vector>

void dl_set_error(const char *, ...) {}

static void pack_elementwise(const PackArgs &a, std::vector<bf16_t> &hi, std::vector<bf16_t> &lo) {
    const size_t total = (size_t)a.rows_pad * a.kstride;
    for (size_t i = 0; i < total; ++i) {
        const int row = (int)(i / a.kstride), k = (int)(i % a.kstride);
        float v = 0.f;
        if (row < a.rows_real) {
            int ph = -1;
            for (int p = 0; p < a.n_phase; ++p)
                if (k >= a.phase_kbase[p] && k < a.phase_kend[p]) ph = p;
            if (ph >= 0) {
                const int kl = k - a.phase_kbase[ph];
                const int tl = kl >> a.log2Cc, c = kl & (a.Cc_pad - 1);
                const int t = a.phase_tap_begin[ph] + tl;
                if (t < a.phase_tap_begin[ph + 1] && c < a.Cc) {
                    const int kh = a.tap_kh[t];
                    int kw = a.tap_kw[t], r = row;
                    if (a.stack_kw) { kw = row % a.KW; r = row / a.KW; }
                    const int ia = a.row_is_a ? r : c, ib = a.row_is_a ? c : r;
                    v = a.src[(((size_t)ia * a.B + ib) * a.KH + kh) * a.KW + kw];
                }
            }
        }
        const bf16_t h = f32_to_bf16(v);
        hi[i] = h;
        lo[i] = f32_to_bf16(v - bf16_to_f32(h));
    }
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<dl_pack_desc> descs;
    dl_pack_desc d;
    while (fread(&d, sizeof(d), 1, f) == 1) descs.push_back(d);
    fclose(f);
    int bad = 0, tiled = 0;
    uint32_t seed = 12345u;
    for (size_t j = 0; j < descs.size(); ++j) {
        const dl_pack_desc &dd = descs[j];
        std::vector<float> src((size_t)dd.A * dd.B * dd.KH * dd.KW);
        for (float &x : src) { seed = seed * 1664525u + 1013904223u; x = ((int)(seed >> 8) - (1 << 23)) * (1.f / (1 << 25)); }
        const size_t total = (size_t)dd.rows_pad * dd.kstride;
        std::vector<bf16_t> hi(total, 0x7fc0), lo(total, 0x7fc0), rhi(total), rlo(total);
        PackArgs a;
        if (const char *why = pack_args_from_desc(&dd, src.data(), hi.data(), lo.data(), a)) { printf("%zu bad descriptor: %s\n", j, why); ++bad; continue; }
        if (!pack_tiled_ok(a)) { printf("%zu chunk form (rows %d x k %d, Cc %d/%d, %dx%d, %d phases, stack %d)\n", j, a.rows_pad, a.kstride, a.Cc, a.Cc_pad, a.KH, a.KW, a.n_phase, a.stack_kw); continue; }
        ++tiled;
        pack_elementwise(a, rhi, rlo);
        std::vector<float> lds(PT_LDS_FLOATS);
        const long tiles = pack_tile_count(a);
        for (long t = 0; t < tiles; ++t) {
            for (float &x : lds) x = -777.f;                 // stale values must never reach the image
            for (int tid = 0; tid < PT_THREADS; ++tid) pack_tile_load(a, (int)t, tid, lds.data());
            for (int tid = 0; tid < PT_THREADS; ++tid) pack_tile_store(a, (int)t, tid, lds.data());
        }
        size_t diff = 0;
        for (size_t i = 0; i < total; ++i) diff += (hi[i] != rhi[i]) + (lo[i] != rlo[i]);
        printf("%zu tiled (rows %d x k %d, Cc %d, %dx%d, %d phases, row_is_a %d, %ld tiles): %zu differing values\n", j, a.rows_pad, a.kstride, a.Cc, a.KH, a.KW,
               a.n_phase, a.row_is_a, tiles, diff);
        if (diff) ++bad;
    }
    printf("%d tiled, %d bad\n", tiled, bad);
    return bad ? 1 : 0;
}
