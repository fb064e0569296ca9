// TEST INFRASTRUCTURE (built by tests/test_jpeg_host.py with g++): runs the per-block / per-pixel arithmetic of the JPEG
// reconstruction kernels - the very header the HIP kernels include, csrc/jpeg_core.h - on the CPU, with loops in place of the
// launch grid, so that it can be compared with oracle/jpeg_baseline.py and with Pillow's libjpeg-turbo without a GPU.
//   probe <info.bin (cris_jpeg_info)> <coef.bin (int16)> <out.rgb (H*W*3 bytes)>
#define CRIS_HD inline
#include "../cris/pytorch_amd/csrc/jpeg_core.h"
#include "../include/cris_hip.h"
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    cris_jpeg_info I;
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(&I, sizeof(I), 1, f) != 1) return 3;
    fclose(f);
    std::vector<short> coef(I.coef_count);
    f = fopen(argv[2], "rb");
    if (!f || fread(coef.data(), 2, coef.size(), f) != coef.size()) return 4;
    fclose(f);
    std::vector<unsigned char> planes(I.plane_bytes), rgb((size_t)I.width * I.height * 3);
    for (int c = 0; c < I.ncomp; ++c) {
        const int bw = I.blocks_w[c];
        for (int b = 0; b < bw * I.blocks_h[c]; ++b) {
            const short* cf = coef.data() + I.coef_offset[c] + (long)b * 64;
            int ws[64];
            for (int col = 0; col < 8; ++col) {
                int in[8], out[8];
                for (int r = 0; r < 8; ++r) in[r] = (int)cf[r * 8 + col] * (int)I.quant[c][r * 8 + col];
                cris_jpeg::idct8<11>(in, out);
                for (int r = 0; r < 8; ++r) ws[r * 8 + col] = out[r];
            }
            const int by = b / bw, bx = b - by * bw;
            for (int r = 0; r < 8; ++r) {
                int out[8];
                cris_jpeg::idct8<18>(ws + r * 8, out);
                for (int col = 0; col < 8; ++col)
                    planes[I.plane_offset[c] + (long)(by * 8 + r) * (bw * 8) + bx * 8 + col] = cris_jpeg::idct_range_limit(out[col]);
            }
        }
    }
    for (int y = 0; y < I.height; ++y)
        for (int x = 0; x < I.width; ++x) {
            unsigned char* o = rgb.data() + ((size_t)y * I.width + x) * 3;
            const int lum = planes[I.plane_offset[0] + (long)y * (I.blocks_w[0] * 8) + x];
            if (I.ncomp == 1) { o[0] = o[1] = o[2] = (unsigned char)lum; continue; }
            const int cb = cris_jpeg::chroma_at(planes.data() + I.plane_offset[1], I.blocks_w[1] * 8, I.down_w[1], I.down_h[1], I.hmax, I.vmax, x, y);
            const int cr = cris_jpeg::chroma_at(planes.data() + I.plane_offset[2], I.blocks_w[2] * 8, I.down_w[2], I.down_h[2], I.hmax, I.vmax, x, y);
            cris_jpeg::ycc_to_rgb(lum, cb, cr, o);
        }
    f = fopen(argv[3], "wb");
    if (!f || fwrite(rgb.data(), 1, rgb.size(), f) != rgb.size()) return 5;
    fclose(f);
    return 0;
}
