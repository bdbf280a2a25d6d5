// Mutation fuzz of the native loaders (tr_obj_parse, tr_image_load_rgba8: host code only) under AddressSanitizer / UBSan.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -o /tmp/loader_fuzz tests/manual/loader_fuzz.cpp tinyrenderer_amd/csrc/tr_loaders.cpp -lz
//   /tmp/loader_fuzz <iterations> <seed> <small .obj / .png / .tga seed files ...>
// Every mutant must be refused or parsed into something consistent (indices in range); no sanitizer report.
#include "../../include/tinyrenderer_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
static std::vector<unsigned char> slurp(const char* p) {
    std::vector<unsigned char> v; FILE* f = std::fopen(p, "rb"); if (!f) return v;
    std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET); v.resize((size_t)n);
    if (std::fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear(); std::fclose(f); return v;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 1000;
    std::mt19937_64 rng(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1);
    std::vector<std::vector<unsigned char>> objs, imgs;
    for (int i = 3; i < argc; ++i) {
        std::string p = argv[i];
        auto v = slurp(argv[i]);
        if (v.empty()) continue;
        if (p.size() > 4 && p.substr(p.size() - 4) == ".obj") { if (v.size() > 20000) v.resize(20000); objs.push_back(v); } else imgs.push_back(v);
    }
    long ok_obj = 0, bad_obj = 0, ok_img = 0, bad_img = 0;
    for (int it = 0; it < iters; ++it) {
        if (!objs.empty()) {
            auto v = objs[rng() % objs.size()];
            int nm = 1 + (int)(rng() % 8);
            for (int m = 0; m < nm; ++m) {
                size_t pos = rng() % v.size();
                switch (rng() % 5) {
                    case 0: v[pos] = (unsigned char)(rng() & 0xFF); break;
                    case 1: v.erase(v.begin() + pos, v.begin() + std::min(v.size(), pos + 1 + rng() % 40)); break;
                    case 2: { const char* t[] = {"f 1/2/3 4//5 6", "v 1e999 -nan inf", "f -1 -2 -3 -4 -5", "f 99999999999 1 2", "vt", "f 1/ /2 3//", "v 1 2", "f 0 0 0", "\r\n", "f 1 2 3 4 5 6 7 8 9 10"}; std::string sx = t[rng() % 10]; sx += "\n"; v.insert(v.begin() + pos, sx.begin(), sx.end()); } break;
                    case 3: v.resize(pos + 1); break;
                    default: v[pos] = "0123456789-/ .\nfvtn"[rng() % 19]; break;
                }
                if (v.empty()) v.push_back('v');
            }
            tr_obj* o = nullptr;
            int rc = tr_obj_parse((const char*)v.data(), (int64_t)v.size(), &o);
            if (rc == 0 && o) {
                int64_t np, nt, nn, ntri; tr_obj_counts(o, &np, &nt, &nn, &ntri);
                const uint32_t* ix = tr_obj_indices(o);
                for (int64_t k = 0; k < ntri * 9; ++k) {   // every index must be in range or None
                    uint32_t a = ix[k]; int64_t lim = (k % 3 == 0) ? np : (k % 3 == 1) ? nt : nn;
                    if (a != 0xFFFFFFFFu && (int64_t)a >= lim) { std::printf("index out of range after a successful parse\n"); return 2; }
                    if (k % 3 == 0 && a == 0xFFFFFFFFu) { std::printf("position index None after a successful parse\n"); return 2; }
                }
                tr_obj_free(o); ++ok_obj;
            } else ++bad_obj;
        }
        if (!imgs.empty()) {
            auto v = imgs[rng() % imgs.size()];
            int nm = 1 + (int)(rng() % 6);
            for (int m = 0; m < nm; ++m) {
                size_t pos = rng() % v.size();
                switch (rng() % 4) {
                    case 0: v[pos] = (unsigned char)(rng() & 0xFF); break;
                    case 1: v[pos] ^= (unsigned char)(1u << (rng() % 8)); break;
                    case 2: v.resize(pos + 1); break;
                    default: if (pos < 64) v[pos] = (unsigned char)(rng() & 0xFF); else v[rng() % 64] = (unsigned char)(rng() & 0xFF); break;
                }
            }
            const char* path = (rng() & 1) ? "/tmp/tr_loader_fuzz.png" : "/tmp/tr_loader_fuzz.tga";
            FILE* f = std::fopen(path, "wb"); std::fwrite(v.data(), 1, v.size(), f); std::fclose(f);
            tr_image* im = nullptr;
            int rc = tr_image_load_rgba8(path, (int)(rng() & 1), &im);
            if (rc == 0 && im) {
                int w, h; tr_image_size(im, &w, &h); const uint8_t* d = tr_image_data(im);
                unsigned long long s = 0; for (size_t k = 0; k < (size_t)w * h * 4; k += 997) s += d[k];  // touch the buffer
                if (s == 0xFFFFFFFFFFFFFFFFull) std::printf("!");
                tr_image_free(im); ++ok_img;
            } else ++bad_img;
        }
    }
    std::printf("obj: %ld parsed, %ld refused; images: %ld decoded, %ld refused\n", ok_obj, bad_obj, ok_img, bad_img);
    return 0;
}
