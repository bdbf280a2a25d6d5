// Native loading surface of the reference's `helpers` module (host only, no GPU needed):
//   load_model                       src/helpers/mod.rs:352-355 (wavefront_obj::obj::parse, first object,
//                                    the triangles of geometry[0] as consumed at model_renderer/mod.rs:237-276)
//   load_rgba_img_from_file          src/helpers/mod.rs:93-96   (image::open(..).into_rgba8())
//   flip_vertically                  src/helpers/mod.rs:291-302
// PNG (8-bit grey / grey+alpha / RGB / RGBA / palette, non-interlaced) is decoded with zlib;
// TGA (BASELINE.json north_star names it; the reference's `image` crate decodes it too) supports
// uncompressed and RLE true-colour / grey images.  Expansion to RGBA8 follows `into_rgba8`:
// L -> (l,l,l,255), LA -> (l,l,l,a), RGB -> a = 255.
#include "../../include/tinyrenderer_hip.h"

#include <zlib.h>

#include <cerrno>
#include <charconv>
#include <new>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct tr_obj {
    std::vector<double> pos, uvw, nrm;
    std::vector<uint32_t> idx;  // 9 per triangle: (v, vt, vn) x 3, 0xFFFFFFFF = None
    std::string name;
};

struct tr_image {
    int w = 0, h = 0;
    std::vector<uint8_t> rgba;  // row-major, row 0 = top (or bottom when loaded flipped)
};

namespace {

thread_local std::string g_load_error;

int load_fail(int code, const std::string& msg) {
    g_load_error = msg;
    return code;
}

bool read_file(const char* path, std::vector<uint8_t>& out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) { std::fclose(f); return false; }
    out.resize((size_t)n);
    size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    return got == (size_t)n;
}

// ------------------------------------------------------------------ OBJ
struct Cursor {
    const char* p;
    const char* end;
    void skip_ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p; }
    bool at_eol() const { return p >= end || *p == '\n' || *p == '#'; }
};

bool parse_double(Cursor& c, double& out) {
    c.skip_ws();
    if (c.at_eol()) return false;
    // the whole token, whatever its length, parsed in place: correctly rounded like Rust's f64 parse and
    // independent of the process locale (strtod follows LC_NUMERIC: a comma-decimal locale would misread
    // every vertex)
    const char* b = c.p;
    while (c.p < c.end && *c.p != ' ' && *c.p != '\t' && *c.p != '\r' && *c.p != '\n' && *c.p != '#') ++c.p;
    const char* e = c.p;
    if (b < e && *b == '+') ++b;  // (from_chars takes no leading plus sign)
    if (b == e || *b == '+') return false;
    const std::from_chars_result r = std::from_chars(b, e, out, std::chars_format::general);
    if (r.ec == std::errc::invalid_argument || r.ptr != e) return false;
    if (r.ec == std::errc::result_out_of_range) {
        // from_chars leaves `out` alone; Rust's parse gives +-inf for too large and +-0 for too small a magnitude:
        // the sign of the decimal exponent tells the two apart
        const bool neg = *b == '-';
        // decimal magnitude = position of the first non-zero digit relative to the point + the exponent field
        long long mag = 0, expo = 0;
        bool seen_point = false, seen_nonzero = false;
        const char* q = b + ((*b == '-') ? 1 : 0);
        for (; q < e && *q != 'e' && *q != 'E'; ++q) {
            if (*q == '.') { seen_point = true; continue; }
            if (*q != '0') seen_nonzero = true;
            if (!seen_point && seen_nonzero) ++mag;       // integer digits from the first non-zero one
            if (seen_point && !seen_nonzero) --mag;        // zeros between the point and the first non-zero digit
        }
        if (q < e) {
            const std::from_chars_result re = std::from_chars(q + 1 + ((q + 1 < e && q[1] == '+') ? 1 : 0), e, expo);
            if (re.ec == std::errc::result_out_of_range) expo = (q + 1 < e && q[1] == '-') ? -(1ll << 40) : (1ll << 40);
        }
        const bool tiny = mag + expo < 0;
        out = tiny ? 0.0 : __builtin_huge_val();
        if (neg) out = -out;
    }
    return true;
}

bool parse_index_triple(Cursor& c, long long& v, long long& t, long long& n, bool& has_t, bool& has_n) {
    c.skip_ws();
    if (c.at_eol()) return false;
    auto read_int = [&](long long& out) -> bool {
        bool neg = false;
        if (c.p < c.end && (*c.p == '-' || *c.p == '+')) { neg = *c.p == '-'; ++c.p; }
        if (c.p >= c.end || *c.p < '0' || *c.p > '9') return false;
        long long x = 0;
        while (c.p < c.end && *c.p >= '0' && *c.p <= '9') { x = x * 10 + (*c.p - '0'); ++c.p; }
        out = neg ? -x : x;
        return true;
    };
    has_t = has_n = false;
    if (!read_int(v)) return false;
    if (c.p < c.end && *c.p == '/') {
        ++c.p;
        if (c.p < c.end && *c.p != '/') {
            if (!read_int(t)) return false;
            has_t = true;
        }
        if (c.p < c.end && *c.p == '/') {
            ++c.p;
            if (!read_int(n)) return false;
            has_n = true;
        }
    }
    return true;
}

int parse_obj_text(const char* text, size_t len, tr_obj* o) {
    Cursor c{text, text + len};
    bool seen_object = false, geometry_closed = false;
    size_t line_no = 0;
    while (c.p < c.end) {
        ++line_no;
        const char* eol = (const char*)std::memchr(c.p, '\n', (size_t)(c.end - c.p));
        if (!eol) eol = c.end;
        Cursor l{c.p, eol};
        c.p = eol < c.end ? eol + 1 : c.end;
        l.skip_ws();
        if (l.at_eol()) continue;
        const char* tag = l.p;
        while (l.p < l.end && *l.p != ' ' && *l.p != '\t' && *l.p != '\r') ++l.p;
        size_t tl = (size_t)(l.p - tag);
        auto is = [&](const char* s) { return std::strlen(s) == tl && std::memcmp(tag, s, tl) == 0; };
        if (is("o")) {
            if (seen_object || !o->pos.empty() || !o->idx.empty()) break;  // only the first object is kept
            seen_object = true;
            l.skip_ws();
            const char* e = l.end;
            while (e > l.p && (e[-1] == '\r' || e[-1] == ' ')) --e;
            o->name.assign(l.p, e);
        } else if (is("v") || is("vn")) {
            double x, y, z;
            if (!parse_double(l, x) || !parse_double(l, y) || !parse_double(l, z))
                return load_fail(TR_ERR_INVALID, "OBJ line " + std::to_string(line_no) + ": bad vertex");
            std::vector<double>& dst = is("v") ? o->pos : o->nrm;
            dst.push_back(x); dst.push_back(y); dst.push_back(z);
        } else if (is("vt")) {
            double u, v = 0.0, w = 0.0;
            if (!parse_double(l, u)) return load_fail(TR_ERR_INVALID, "OBJ line " + std::to_string(line_no) + ": bad vt");
            if (parse_double(l, v)) parse_double(l, w);
            o->uvw.push_back(u); o->uvw.push_back(v); o->uvw.push_back(w);
        } else if (is("usemtl")) {
            if (!o->idx.empty()) geometry_closed = true;  // later faces belong to geometry[1..]
        } else if (is("f") && !geometry_closed) {
            std::vector<uint32_t> corners;  // 3 per corner; a polygon may have any number of corners
            int nc = 0;
            long long v, t = 0, n = 0;
            bool ht, hn;
            while (parse_index_triple(l, v, t, n, ht, hn)) {
                long long nv = (long long)(o->pos.size() / 3), nt = (long long)(o->uvw.size() / 3), nn = (long long)(o->nrm.size() / 3);
                long long vi = v > 0 ? v - 1 : nv + v;
                long long ti = ht ? (t > 0 ? t - 1 : nt + t) : -1;
                long long ni = hn ? (n > 0 ? n - 1 : nn + n) : -1;
                if (vi < 0 || vi >= nv || (ht && (ti < 0 || ti >= nt)) || (hn && (ni < 0 || ni >= nn)))
                    return load_fail(TR_ERR_INVALID, "OBJ line " + std::to_string(line_no) + ": index out of range");
                corners.push_back((uint32_t)vi);
                corners.push_back(ht ? (uint32_t)ti : 0xFFFFFFFFu);
                corners.push_back(hn ? (uint32_t)ni : 0xFFFFFFFFu);
                ++nc;
            }
            l.skip_ws();
            if (!l.at_eol()) return load_fail(TR_ERR_INVALID, "OBJ line " + std::to_string(line_no) + ": bad face corner");
            for (int k = 1; k + 1 < nc; ++k) {  // polygon -> triangle fan
                const int tri[3] = {0, k, k + 1};
                for (int q = 0; q < 3; ++q)
                    for (int d = 0; d < 3; ++d) o->idx.push_back(corners[3 * (size_t)tri[q] + d]);
            }
        }
    }
    return TR_OK;
}

// ------------------------------------------------------------------ PNG
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int decode_png(const std::vector<uint8_t>& f, tr_image* img) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (f.size() < 8 || std::memcmp(f.data(), sig, 8) != 0) return load_fail(TR_ERR_INVALID, "not a PNG");
    size_t at = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool end = false;
    while (!end && at + 12 <= f.size()) {
        uint32_t len = be32(&f[at]);
        const uint8_t* type = &f[at + 4];
        if (at + 12 + (size_t)len > f.size()) return load_fail(TR_ERR_INVALID, "PNG chunk runs past the end");
        const uint8_t* data = &f[at + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        at += 12 + (size_t)len;
    }
    if (!w || !h || w > 32768 || h > 32768) return load_fail(TR_ERR_INVALID, "PNG: bad dimensions");
    if (depth != 8) return load_fail(TR_ERR_UNSUPPORTED, "PNG: only 8-bit samples are supported");
    if (interlace) return load_fail(TR_ERR_UNSUPPORTED, "PNG: interlaced images are not supported");
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) return load_fail(TR_ERR_INVALID, "PNG: bad colour type");
    if (ctype == 3 && plte.empty()) return load_fail(TR_ERR_INVALID, "PNG: palette missing");
    const size_t stride = (size_t)w * ch;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || raw_len != raw.size()) return load_fail(TR_ERR_INVALID, "PNG: zlib stream is corrupt");
    // unfilter in place (PNG spec section 9)
    std::vector<uint8_t> pix(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t* src = &raw[(stride + 1) * y + 1];
        uint8_t* cur = &pix[stride * y];
        const uint8_t* up = y ? &pix[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= (size_t)ch ? cur[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return load_fail(TR_ERR_INVALID, "PNG: bad filter type");
            }
            cur[x] = (uint8_t)(src[x] + pred);
        }
    }
    img->w = (int)w;
    img->h = (int)h;
    img->rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t* s = &pix[i * ch];
        uint8_t* d = &img->rgba[i * 4];
        switch (ctype) {
            case 0: d[0] = d[1] = d[2] = s[0]; d[3] = 255; break;
            case 2: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 255; break;
            case 3: {
                size_t k = s[0];
                if (k * 3 + 2 >= plte.size()) return load_fail(TR_ERR_INVALID, "PNG: palette index out of range");
                d[0] = plte[k * 3]; d[1] = plte[k * 3 + 1]; d[2] = plte[k * 3 + 2];
                d[3] = k < trns.size() ? trns[k] : 255;
                break;
            }
            case 4: d[0] = d[1] = d[2] = s[0]; d[3] = s[1]; break;
            case 6: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3]; break;
        }
    }
    return TR_OK;
}

// ------------------------------------------------------------------ TGA
int decode_tga(const std::vector<uint8_t>& f, tr_image* img) {
    if (f.size() < 18) return load_fail(TR_ERR_INVALID, "TGA: file too short");
    const int id_len = f[0], cmap_type = f[1], itype = f[2];
    const int cmap_len = f[5] | (f[6] << 8), cmap_bits = f[7];
    const int w = f[12] | (f[13] << 8), h = f[14] | (f[15] << 8), bpp = f[16], desc = f[17];
    if (w <= 0 || h <= 0) return load_fail(TR_ERR_INVALID, "TGA: bad dimensions");
    const bool rle = itype == 10 || itype == 11;
    const bool grey = itype == 3 || itype == 11;
    if (!(itype == 2 || itype == 3 || itype == 10 || itype == 11)) return load_fail(TR_ERR_UNSUPPORTED, "TGA: only true-colour / grey images (types 2, 3, 10, 11)");
    if (!((grey && bpp == 8) || (!grey && (bpp == 24 || bpp == 32)))) return load_fail(TR_ERR_UNSUPPORTED, "TGA: unsupported pixel depth");
    size_t at = 18 + (size_t)id_len + (cmap_type ? (size_t)cmap_len * ((cmap_bits + 7) / 8) : 0);
    const int bytes = bpp / 8;
    std::vector<uint8_t> pix((size_t)w * h * bytes);
    if (!rle) {
        if (at + pix.size() > f.size()) return load_fail(TR_ERR_INVALID, "TGA: pixel data truncated");
        std::memcpy(pix.data(), &f[at], pix.size());
    } else {
        size_t out = 0;
        while (out < pix.size()) {
            if (at >= f.size()) return load_fail(TR_ERR_INVALID, "TGA: RLE data truncated");
            const int hdr = f[at++];
            const size_t count = (size_t)(hdr & 0x7F) + 1;
            if (hdr & 0x80) {
                if (at + bytes > f.size()) return load_fail(TR_ERR_INVALID, "TGA: RLE data truncated");
                for (size_t k = 0; k < count && out < pix.size(); ++k, out += bytes) std::memcpy(&pix[out], &f[at], bytes);
                at += bytes;
            } else {
                size_t n = count * bytes;
                if (at + n > f.size() || out + n > pix.size()) return load_fail(TR_ERR_INVALID, "TGA: RLE data truncated");
                std::memcpy(&pix[out], &f[at], n);
                at += n;
                out += n;
            }
        }
    }
    const bool top_origin = (desc & 0x20) != 0, right_origin = (desc & 0x10) != 0;
    img->w = w;
    img->h = h;
    img->rgba.resize((size_t)w * h * 4);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int sy = top_origin ? y : h - 1 - y, sx = right_origin ? w - 1 - x : x;
            const uint8_t* s = &pix[((size_t)sy * w + sx) * bytes];
            uint8_t* d = &img->rgba[((size_t)y * w + x) * 4];
            if (grey) { d[0] = d[1] = d[2] = s[0]; d[3] = 255; }
            else { d[0] = s[2]; d[1] = s[1]; d[2] = s[0]; d[3] = bytes == 4 ? s[3] : 255; }  // BGR(A) on disk
        }
    return TR_OK;
}

}  // namespace

extern "C" {

const char* tr_load_error(void) { return g_load_error.c_str(); }

// No exception may unwind through the C boundary: an allocation that fails inside a loader comes back as
// TR_ERR_OOM, anything else as TR_ERR_INVALID.
#define TR_GUARDED(...)                                                      \
    try { __VA_ARGS__ }                                                      \
    catch (const std::bad_alloc&) { return TR_ERR_OOM; }                     \
    catch (...) { return TR_ERR_INVALID; }

int tr_obj_parse(const char* text, int64_t len, tr_obj** out) {
    if (!text || len < 0 || !out) return TR_ERR_INVALID;
    *out = nullptr;
    TR_GUARDED(
        tr_obj* o = new (std::nothrow) tr_obj();
        if (!o) return TR_ERR_OOM;
        int rc;
        try { rc = parse_obj_text(text, (size_t)len, o); }
        catch (...) { delete o; throw; }
        if (rc != TR_OK) { delete o; return rc; }
        *out = o;
        return TR_OK;
    )
}

int tr_obj_load(const char* path, tr_obj** out) {
    if (!path || !out) return TR_ERR_INVALID;
    *out = nullptr;
    TR_GUARDED(
        std::vector<uint8_t> f;
        if (!read_file(path, f)) return load_fail(TR_ERR_INVALID, std::string("cannot read ") + path + ": " + std::strerror(errno));
        return tr_obj_parse((const char*)f.data(), (int64_t)f.size(), out);
    )
}

int tr_obj_counts(const tr_obj* o, int64_t* n_positions, int64_t* n_uvw, int64_t* n_normals, int64_t* n_triangles) {
    if (!o) return TR_ERR_INVALID;
    if (n_positions) *n_positions = (int64_t)(o->pos.size() / 3);
    if (n_uvw) *n_uvw = (int64_t)(o->uvw.size() / 3);
    if (n_normals) *n_normals = (int64_t)(o->nrm.size() / 3);
    if (n_triangles) *n_triangles = (int64_t)(o->idx.size() / 9);
    return TR_OK;
}
const double* tr_obj_positions(const tr_obj* o) { return o ? o->pos.data() : nullptr; }
const double* tr_obj_uvw(const tr_obj* o) { return o ? o->uvw.data() : nullptr; }
const double* tr_obj_normals(const tr_obj* o) { return o ? o->nrm.data() : nullptr; }
const uint32_t* tr_obj_indices(const tr_obj* o) { return o ? o->idx.data() : nullptr; }
void tr_obj_free(tr_obj* o) { delete o; }

int tr_image_load_rgba8(const char* path, int flip_vertically, tr_image** out) {
    if (!path || !out) return TR_ERR_INVALID;
    *out = nullptr;
    tr_image* img = nullptr;
    try {
        std::vector<uint8_t> f;
        if (!read_file(path, f)) return load_fail(TR_ERR_INVALID, std::string("cannot read ") + path + ": " + std::strerror(errno));
        img = new (std::nothrow) tr_image();
        if (!img) return TR_ERR_OOM;
        int rc;
        if (f.size() >= 8 && f[0] == 0x89 && f[1] == 'P') rc = decode_png(f, img);
        else {
            const char* dot = std::strrchr(path, '.');
            if (dot && (!strcasecmp(dot, ".tga") || !strcasecmp(dot, ".icb") || !strcasecmp(dot, ".vda") || !strcasecmp(dot, ".vst"))) rc = decode_tga(f, img);
            else rc = load_fail(TR_ERR_UNSUPPORTED, "unsupported image format (PNG and TGA are decoded)");
        }
        if (rc != TR_OK) { delete img; return rc; }
        if (flip_vertically) {  // src/helpers/mod.rs:291-302
            const size_t row = (size_t)img->w * 4;
            std::vector<uint8_t> tmp(row);
            for (int y = 0; y < img->h / 2; ++y) {
                uint8_t* a = &img->rgba[row * y];
                uint8_t* b = &img->rgba[row * (size_t)(img->h - 1 - y)];
                std::memcpy(tmp.data(), a, row);
                std::memcpy(a, b, row);
                std::memcpy(b, tmp.data(), row);
            }
        }
        *out = img;
        return TR_OK;
    } catch (const std::bad_alloc&) {
        delete img;
        return TR_ERR_OOM;
    } catch (...) {
        delete img;
        return TR_ERR_INVALID;
    }
}
int tr_image_size(const tr_image* img, int* width, int* height) {
    if (!img) return TR_ERR_INVALID;
    if (width) *width = img->w;
    if (height) *height = img->h;
    return TR_OK;
}
const uint8_t* tr_image_data(const tr_image* img) { return img ? img->rgba.data() : nullptr; }
void tr_image_free(tr_image* img) { delete img; }

// helpers::_write_nd_rgba_img_to_file (src/helpers/mod.rs:114-120): 8-bit RGBA PNG, rows as given
// (row 0 = top), filter 0 on every row, one zlib stream in one IDAT.
int tr_image_write_png(const char* path, const uint8_t* rgba8, int width, int height) {
    if (!path || !rgba8 || width < 1 || height < 1) return TR_ERR_INVALID;
    TR_GUARDED(
    const size_t row = (size_t)width * 4;
    std::vector<uint8_t> raw((row + 1) * (size_t)height);
    for (int y = 0; y < height; ++y) {
        raw[(row + 1) * y] = 0;
        std::memcpy(&raw[(row + 1) * y + 1], rgba8 + row * y, row);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return load_fail(TR_ERR_HIP, "zlib compress failed");
    FILE* f = std::fopen(path, "wb");
    if (!f) return load_fail(TR_ERR_INVALID, std::string("cannot write ") + path + ": " + std::strerror(errno));
    auto put32 = [](uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
    bool ok = true;
    auto chunk = [&](const char* type, const uint8_t* data, size_t len) {
        uint8_t hdr[8];
        put32(hdr, (uint32_t)len);
        std::memcpy(hdr + 4, type, 4);
        uLong crc = crc32(0L, hdr + 4, 4);
        if (len) crc = crc32(crc, data, (uInt)len);
        uint8_t tail[4];
        put32(tail, (uint32_t)crc);
        ok = ok && std::fwrite(hdr, 1, 8, f) == 8 && (len == 0 || std::fwrite(data, 1, len, f) == len) && std::fwrite(tail, 1, 4, f) == 4;
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    ok = std::fwrite(sig, 1, 8, f) == 8;
    uint8_t ihdr[13];
    put32(ihdr, (uint32_t)width);
    put32(ihdr + 4, (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;  // 8-bit RGBA, no interlace
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (size_t)zlen);
    chunk("IEND", nullptr, 0);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? TR_OK : load_fail(TR_ERR_INVALID, std::string("short write to ") + path);
    )
}

}  // extern "C"
