// f3ds_host.cpp -- host-side pieces of libf3ds that need no GPU: parameter defaults, error
// strings, PCD v0.7 reader/writer (the step either side of the hot path: replaces
// pcl::io::loadPCDFile at /root/reference/src/supervoxel_clustering.cpp:313 and adds the
// coloured-cloud writer the reference lacks), the deterministic synthetic-frame generators
// of BASELINE.md section 4, and the label colour table.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/f3ds.h"
#include "f3ds_glasbey.h"
#include "f3ds_build_stamp.h"
#include "f3ds_math.h"
#include "f3ds_dev.h"

extern "C" {

void f3ds_default_params(f3ds_params* p) {
    if (!p) return;
    // defaults of /root/reference/src/supervoxel_clustering.cpp:247-267 and Clustering()
    // (src/clustering.cpp:533-539); merging defaults to --AL (:275-279)
    p->voxel_res = 0.008f;
    p->seed_res = 0.08f;
    p->w_color = 0.2f;
    p->w_spatial = 0.4f;
    p->w_normal = 1.0f;
    p->use_transform = 1;
    p->color_metric = F3DS_LAB_CIEDE00;
    p->geom_metric = F3DS_NORMALS_DIFF;
    p->merging = F3DS_ADAPTIVE_LAMBDA;
    p->lambda = 0.0f;
    p->bins = 0;
    p->threshold = 0.0f;
    p->leaf_order = 0;
    p->fold_negative_z = 1;
}

int f3ds_version(void) { return F3DS_VERSION; }
// "f3ds <version> src:<stamp>[ +whatif][ +dev]": the stamp is the hash of the sources this library was built from (csrc/Makefile);
// " +dev" = the process has F3DS_DEV set, i.e. the development switches of csrc/f3ds_dev.h are being read (bench.py then reports no value)
const char* f3ds_version_string(void) {
#ifdef F3DS_WHATIF
#define F3DS_VERSION_TEXT "f3ds 1.2.0 src:" F3DS_BUILD_STAMP " +whatif"
#else
#define F3DS_VERSION_TEXT "f3ds 1.2.0 src:" F3DS_BUILD_STAMP
#endif
    return f3ds::dev_mode() ? F3DS_VERSION_TEXT " +dev" : F3DS_VERSION_TEXT;
#undef F3DS_VERSION_TEXT
}
int f3ds_dev_mode(void) { return f3ds::dev_mode() ? 1 : 0; }

const char* f3ds_strerror(int code) {
    switch (code) {
        case F3DS_OK: return "ok";
        case F3DS_ERR_ARG: return "invalid argument";
        case F3DS_ERR_NO_DEVICE: return "no HIP device available (libf3ds has no CPU fallback)";
        case F3DS_ERR_HIP: return "HIP runtime call failed";
        case F3DS_ERR_DEPTH: return "voxel grid deeper than 21 octree levels";
        case F3DS_ERR_LOGIC: return "Cannot call 'cluster' before setting an initial state";
        case F3DS_ERR_RANGE: return "Argument outside range";
        case F3DS_ERR_UNSUPPORTED: return "degenerate input not supported by the device path";
        case F3DS_ERR_IO: return "PCD file i/o error";
        case F3DS_ERR_EQ_BIN: return "equalization bin out of range (delta == 1.0 under --EQ)";
        case F3DS_ERR_CAPACITY: return "output buffer too small";
        case F3DS_ERR_BUSY: return "frame pipeline busy";
        case F3DS_ERR_EMPTY: return "frame pipeline empty";
        case F3DS_ERR_OUT_OF_RANGE: return "out of range: unknown supervoxel label in an adjacency, or threshold bounds outside [0, 1]";
    }
    return "unknown error";
}

uint32_t f3ds_label_color(uint32_t label) { return f3ds_glasbey_256[label % 256u]; }

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// PCD v0.7
// ------------------------------------------------------------------------------------------------
namespace {

struct Field { std::string name; int size = 4; char type = 'F'; int count = 1; size_t offset = 0; };

// liblzf stream: control byte c; c < 32 -> c+1 literals; else back-reference of len (c>>5)+2
// (len field 7 -> one more length byte) at distance ((c&31)<<8 | next)+1, overlap allowed.
bool lzf_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len) {
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        unsigned ctrl = in[ip++];
        if (ctrl < 32) {
            ctrl++;
            if (op + ctrl > out_len || ip + ctrl > in_len) return false;
            memcpy(out + op, in + ip, ctrl);
            op += ctrl; ip += ctrl;
        } else {
            unsigned len = ctrl >> 5;
            if (len == 7) { if (ip >= in_len) return false; len += in[ip++]; }
            if (ip >= in_len) return false;
            size_t dist = ((size_t)(ctrl & 31) << 8) + in[ip++] + 1;
            len += 2;
            if (dist > op || op + len > out_len) return false;
            for (unsigned k = 0; k < len; ++k, ++op) out[op] = out[op - dist];
        }
    }
    return op == out_len;
}

struct P16 { float x, y, z; uint32_t rgba; };

}  // namespace

extern "C" int f3ds_pcd_read(const char* path, void* points16, uint32_t* labels, size_t cap, size_t* n_out,
                             uint32_t* width, uint32_t* height) {
    if (!path) return F3DS_ERR_ARG;
    FILE* f = fopen(path, "rb");
    if (!f) return F3DS_ERR_IO;
    std::vector<uint8_t> data;
    {
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz < 0) { fclose(f); return F3DS_ERR_IO; }
        data.resize((size_t)sz);
        if (sz && fread(data.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return F3DS_ERR_IO; }
        fclose(f);
    }
    std::vector<Field> fields;
    size_t npoints = 0, w = 0, h = 0, pos = 0;
    std::string mode;
    bool have_points = false;
    while (pos < data.size()) {
        size_t eol = pos;
        while (eol < data.size() && data[eol] != '\n') eol++;
        std::string line((const char*)data.data() + pos, eol - pos);
        pos = eol + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::vector<std::string> tok;
        {
            size_t i = 0;
            while (i < line.size()) {
                while (i < line.size() && (line[i] == ' ' || line[i] == '\t')) i++;
                size_t j = i;
                while (j < line.size() && line[j] != ' ' && line[j] != '\t') j++;
                if (j > i) tok.push_back(line.substr(i, j - i));
                i = j;
            }
        }
        if (tok.empty()) continue;
        if (tok[0] == "FIELDS" || tok[0] == "COLUMNS") {
            fields.resize(tok.size() - 1);
            for (size_t i = 1; i < tok.size(); ++i) fields[i - 1].name = tok[i];
        } else if (tok[0] == "SIZE") {
            for (size_t i = 1; i < tok.size() && i - 1 < fields.size(); ++i) fields[i - 1].size = atoi(tok[i].c_str());
        } else if (tok[0] == "TYPE") {
            for (size_t i = 1; i < tok.size() && i - 1 < fields.size(); ++i) fields[i - 1].type = tok[i][0];
        } else if (tok[0] == "COUNT") {
            for (size_t i = 1; i < tok.size() && i - 1 < fields.size(); ++i) fields[i - 1].count = atoi(tok[i].c_str());
        } else if (tok[0] == "WIDTH" && tok.size() > 1) {
            w = strtoull(tok[1].c_str(), nullptr, 10);
        } else if (tok[0] == "HEIGHT" && tok.size() > 1) {
            h = strtoull(tok[1].c_str(), nullptr, 10);
        } else if (tok[0] == "POINTS" && tok.size() > 1) {
            npoints = strtoull(tok[1].c_str(), nullptr, 10);
            have_points = true;
        } else if (tok[0] == "DATA" && tok.size() > 1) {
            mode = tok[1];
            break;
        }
    }
    if (mode.empty() || fields.empty()) return F3DS_ERR_IO;
    if (pos > data.size()) pos = data.size();      // "DATA binary" as the last bytes of the file, no newline: no payload
    for (const auto& f : fields) if (f.size <= 0 || f.count <= 0 || f.size > 8 || f.count > (1 << 20)) return F3DS_ERR_IO;
    if (!have_points) { if (h && w > (size_t)-1 / h) return F3DS_ERR_IO; npoints = w * h; }
    if (width) *width = (uint32_t)w;
    if (height) *height = (uint32_t)h;
    if (n_out) *n_out = npoints;
    if (!points16) return F3DS_OK;
    if (cap < npoints) return F3DS_ERR_CAPACITY;
    size_t stride = 0;
    int ix = -1, iy = -1, iz = -1, irgb = -1, ilabel = -1;
    for (size_t i = 0; i < fields.size(); ++i) {
        fields[i].offset = stride;
        stride += (size_t)fields[i].size * (size_t)fields[i].count;
        if (fields[i].name == "x") ix = (int)i;
        else if (fields[i].name == "y") iy = (int)i;
        else if (fields[i].name == "z") iz = (int)i;
        else if (fields[i].name == "rgb" || fields[i].name == "rgba") irgb = (int)i;
        else if (fields[i].name == "label") ilabel = (int)i;
    }
    if (ix < 0 || iy < 0 || iz < 0) return F3DS_ERR_IO;
    for (int i : {ix, iy, iz}) if (fields[i].size != 4 || fields[i].type != 'F') return F3DS_ERR_IO;
    // rgb / label are read as 4-byte words in the binary layouts
    if (mode != "ascii") for (int i : {irgb, ilabel}) if (i >= 0 && fields[i].size * fields[i].count < 4) return F3DS_ERR_IO;
    if (stride == 0 || npoints > (size_t)-1 / stride) return F3DS_ERR_IO;
    P16* out = (P16*)points16;
    auto put = [&](size_t i, const uint8_t* px, const uint8_t* py, const uint8_t* pz, const uint8_t* pc, const uint8_t* pl) {
        memcpy(&out[i].x, px, 4); memcpy(&out[i].y, py, 4); memcpy(&out[i].z, pz, 4);
        uint32_t c = 0;
        if (pc) memcpy(&c, pc, 4);
        out[i].rgba = c;
        if (labels) { uint32_t l = 0; if (pl) memcpy(&l, pl, 4); labels[i] = l; }
    };
    if (mode == "binary") {
        if (npoints > (data.size() - pos) / stride) return F3DS_ERR_IO;
        const uint8_t* base = data.data() + pos;
        for (size_t i = 0; i < npoints; ++i) {
            const uint8_t* r = base + i * stride;
            put(i, r + fields[ix].offset, r + fields[iy].offset, r + fields[iz].offset,
                irgb >= 0 ? r + fields[irgb].offset : nullptr, ilabel >= 0 ? r + fields[ilabel].offset : nullptr);
        }
    } else if (mode == "binary_compressed") {
        if (data.size() - pos < 8) return F3DS_ERR_IO;
        uint32_t csize, usize;
        memcpy(&csize, data.data() + pos, 4); memcpy(&usize, data.data() + pos + 4, 4);
        if (data.size() - pos - 8 < csize || (size_t)usize != stride * npoints) return F3DS_ERR_IO;
        std::vector<uint8_t> buf(usize);
        if (!lzf_decompress(data.data() + pos + 8, csize, buf.data(), usize)) return F3DS_ERR_IO;
        // field-major: all of field 0, then all of field 1, ...
        std::vector<size_t> fbase(fields.size());
        size_t acc = 0;
        for (size_t i = 0; i < fields.size(); ++i) { fbase[i] = acc; acc += (size_t)fields[i].size * fields[i].count * npoints; }
        for (size_t i = 0; i < npoints; ++i) {
            auto at = [&](int fi) { return buf.data() + fbase[fi] + i * (size_t)fields[fi].size * fields[fi].count; };
            put(i, at(ix), at(iy), at(iz), irgb >= 0 ? at(irgb) : nullptr, ilabel >= 0 ? at(ilabel) : nullptr);
        }
    } else if (mode == "ascii") {
        const char* p = (const char*)data.data() + pos;
        const char* end = (const char*)data.data() + data.size();
        std::string s(p, end);
        char* cur = &s[0];
        for (size_t i = 0; i < npoints; ++i) {
            float xyz[3] = {0, 0, 0};
            uint32_t c = 0, l = 0;
            for (size_t fi = 0; fi < fields.size(); ++fi) {
                for (int k = 0; k < fields[fi].count; ++k) {
                    while (*cur == ' ' || *cur == '\t' || *cur == '\n' || *cur == '\r') cur++;
                    if (!*cur) return F3DS_ERR_IO;
                    char* nx = cur;
                    if (fields[fi].type == 'F') {
                        float v = strtof(cur, &nx);           // "nan" parses to NaN
                        if ((int)fi == ix) xyz[0] = v; else if ((int)fi == iy) xyz[1] = v; else if ((int)fi == iz) xyz[2] = v;
                        else if ((int)fi == irgb) memcpy(&c, &v, 4);   // legacy float-packed rgb
                    } else if (fields[fi].type == 'U') {
                        unsigned long long v = strtoull(cur, &nx, 10);
                        if ((int)fi == irgb) c = (uint32_t)v; else if ((int)fi == ilabel) l = (uint32_t)v;
                    } else {
                        long long v = strtoll(cur, &nx, 10);
                        if ((int)fi == irgb) c = (uint32_t)v; else if ((int)fi == ilabel) l = (uint32_t)v;
                    }
                    if (nx == cur) return F3DS_ERR_IO;
                    cur = nx;
                }
            }
            out[i].x = xyz[0]; out[i].y = xyz[1]; out[i].z = xyz[2]; out[i].rgba = c;
            if (labels) labels[i] = l;
        }
    } else {
        return F3DS_ERR_IO;
    }
    return F3DS_OK;
}

extern "C" int f3ds_pcd_write(const char* path, const float* xyz, const uint32_t* rgba, const uint32_t* labels, size_t n,
                              int mode) {
    if (!path || (!xyz && n)) return F3DS_ERR_ARG;
    FILE* f = fopen(path, "wb");
    if (!f) return F3DS_ERR_IO;
    const bool hl = labels != nullptr;
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n");
    fprintf(f, "FIELDS x y z rgba%s\nSIZE 4 4 4 4%s\nTYPE F F F U%s\nCOUNT 1 1 1 1%s\n", hl ? " label" : "", hl ? " 4" : "",
            hl ? " U" : "", hl ? " 1" : "");
    fprintf(f, "WIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA %s\n", n, n, mode == 1 ? "binary" : "ascii");
    for (size_t i = 0; i < n; ++i) {
        uint32_t c = rgba ? rgba[i] : 0u;
        if (mode == 1) {
            fwrite(xyz + 3 * i, 4, 3, f);
            fwrite(&c, 4, 1, f);
            if (hl) fwrite(labels + i, 4, 1, f);
        } else {
            fprintf(f, "%.9g %.9g %.9g %u", xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], c);
            if (hl) fprintf(f, " %u", labels[i]);
            fputc('\n', f);
        }
    }
    bool ok = !ferror(f);
    fclose(f);
    return ok ? F3DS_OK : F3DS_ERR_IO;
}

// ------------------------------------------------------------------------------------------------
// synthetic frames: SplitMix64, every sample keyed by (seed, index) so the
// bytes do not depend on iteration order or thread count.
// ------------------------------------------------------------------------------------------------
namespace {

inline uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
// The generator's own logarithm and cosine: round 1's series (separate multiply and add), frozen here so that the synthetic
// frames of BASELINE.md stay byte-identical while csrc/f3ds_math.h (the arithmetic of the path) is free to change.
inline double gen_estrin8(double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7, double z) {
    const double z2 = z * z, z4 = z2 * z2;
    const double a0 = c0 + c1 * z, a1 = c2 + c3 * z, a2 = c4 + c5 * z, a3 = c6 + c7 * z;
    const double b0 = a0 + a1 * z2, b1 = a2 + a3 * z2;
    return b0 + b1 * z4;
}

inline double gen_estrin16(double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7, double c8, double c9,
                          double c10, double c11, double c12, double c13, double c14, double c15, double z) {
    const double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double a0 = c0 + c1 * z, a1 = c2 + c3 * z, a2 = c4 + c5 * z, a3 = c6 + c7 * z;
    const double a4 = c8 + c9 * z, a5 = c10 + c11 * z, a6 = c12 + c13 * z, a7 = c14 + c15 * z;
    const double b0 = a0 + a1 * z2, b1 = a2 + a3 * z2, b2 = a4 + a5 * z2, b3 = a6 + a7 * z2;
    const double d0 = b0 + b1 * z4, d1 = b2 + b3 * z4;
    return d0 + d1 * z8;
}

inline double gen_log(double x) {
    if (f3ds::m_isnan(x)) return x;
    if (x < 0.0) return f3ds::m_nan();
    if (x == 0.0) return -f3ds::m_inf();
    if (f3ds::m_isinf(x)) return x;
    int e = 0;
    if (x < 0x1p-1022) { x = x * 0x1p54; e = -54; }   // subnormal
    uint64_t u = f3ds::m_bits(x);
    e += (int)(u >> 52) - 1023;
    double m = f3ds::m_from_bits((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);   // [1,2)
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }                       // [0.7071,1.4142]
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    // 2*atanh(s) = 2s * (1 + z/3 + z^2/5 + ...),  z <= 0.0295: 16 terms of 1/(2k+3)
    const double p = gen_estrin16(1.0 / 3.0, 1.0 / 5.0, 1.0 / 7.0, 1.0 / 9.0, 1.0 / 11.0, 1.0 / 13.0, 1.0 / 15.0, 1.0 / 17.0, 1.0 / 19.0, 1.0 / 21.0,
                                1.0 / 23.0, 1.0 / 25.0, 1.0 / 27.0, 1.0 / 29.0, 1.0 / 31.0, 1.0 / 33.0, z);
    const double LN2_HI = 0x1.62e4200000000p-1;
    const double LN2_LO = 0x1.fdf473de6af28p-22;
    double ed = (double)e;
    double two_s = 2.0 * s;
    double r = ed * LN2_LO + two_s * (z * p);
    r = r + two_s;
    r = r + ed * LN2_HI;
    return r;
}

inline double gen_sin_kernel(double r) {
    const double z = r * r;
    const double p = gen_estrin8(-1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0, -1.0 / 39916800.0, 1.0 / 6227020800.0,
                               -1.0 / 1307674368000.0, 1.0 / 355687428096000.0, z);
    return r + r * (z * p);
}

inline double gen_cos_kernel(double r) {
    const double z = r * r;
    const double p = gen_estrin16(-0.5, 1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0, -1.0 / 3628800.0, 1.0 / 479001600.0, -1.0 / 87178291200.0,
                                1.0 / 20922789888000.0, -1.0 / 6402373705728000.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, z);
    return 1.0 + z * p;
}

inline int gen_rem_pio2(double x, double* r) {
    const double TWO_OVER_PI = 0.6366197723675814;
    const double PIO2_1 = 0x1.921fb54400000p+0;      // 33 bits
    const double PIO2_2 = 0x1.0b4611a600000p-34;     // next 33 bits
    const double PIO2_3 = 0x1.3198a2e037073p-69;
    double t = x * TWO_OVER_PI;
    double nd = (double)(long long)(t + (t < 0.0 ? -0.5 : 0.5));
    *r = ((x - nd * PIO2_1) - nd * PIO2_2) - nd * PIO2_3;
    return (int)((long long)nd & 3);
}

inline double gen_cos(double x) {
    if (f3ds::m_isnan(x) || f3ds::m_isinf(x)) return f3ds::m_nan();
    if (f3ds::m_abs(x) <= 0.7853981633974483) return gen_cos_kernel(x);
    if (f3ds::m_abs(x) > 1.0e15) return f3ds::m_nan();
    double r; int q = gen_rem_pio2(x, &r);
    switch (q) {
        case 0: return gen_cos_kernel(r);
        case 1: return -gen_sin_kernel(r);
        case 2: return -gen_cos_kernel(r);
        default: return gen_sin_kernel(r);
    }
}

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    double uni() { return (double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }
    double uni(double a, double b) { return a + (b - a) * uni(); }
    double gauss() {   // Box-Muller on IEEE basic operations (platform independent)
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        double r2 = -2.0 * gen_log(u1);
        double r = r2 > 0 ? __builtin_sqrt(r2) : 0.0;
        return r * gen_cos(6.283185307179586 * u2);
    }
};
struct Box { double lo[3], hi[3]; uint8_t col[3]; };
struct Sphere { double c[3], r; uint8_t col[3]; };

inline uint8_t jitter_col(int base, Rng& g, int amp) {
    int v = base + (int)(g.uni() * (2 * amp + 1)) - amp;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

void synth_rgbd(uint64_t seed, uint32_t W, uint32_t H, uint32_t nan_permille, P16* out) {
    Rng sg(seed * 0x2545F4914F6CDD1DULL + 17);
    const double f = (W == 640 ? 525.0 : 0.8 * (double)W);
    const double cx = 0.5 * ((double)W - 1.0), cy = 0.5 * ((double)H - 1.0);
    std::vector<Box> boxes(12);
    std::vector<Sphere> spheres(6);
    for (Box& b : boxes) {
        double c[3] = {sg.uni(-1.2, 1.2), sg.uni(-0.2, 0.9), sg.uni(0.8, 2.8)};
        double hsz[3] = {sg.uni(0.08, 0.3), sg.uni(0.08, 0.3), sg.uni(0.08, 0.3)};
        for (int a = 0; a < 3; ++a) { b.lo[a] = c[a] - hsz[a]; b.hi[a] = c[a] + hsz[a]; }
        if (b.hi[1] > 1.0) { b.lo[1] -= b.hi[1] - 1.0; b.hi[1] = 1.0; }   // rest on the floor
        for (int a = 0; a < 3; ++a) b.col[a] = (uint8_t)(30 + (int)(sg.uni() * 200));
    }
    for (Sphere& s : spheres) {
        s.c[0] = sg.uni(-1.0, 1.0); s.c[1] = sg.uni(-0.4, 0.7); s.c[2] = sg.uni(0.9, 2.6);
        s.r = sg.uni(0.08, 0.22);
        for (int a = 0; a < 3; ++a) s.col[a] = (uint8_t)(30 + (int)(sg.uni() * 200));
    }
    const uint8_t wall_back[3] = {200, 190, 170}, wall_left[3] = {150, 170, 200}, floor_c[3] = {120, 100, 80};
    const float qnan = f3ds::m_from_bitsf(0x7fc00000u);
    for (uint32_t v = 0; v < H; ++v)
        for (uint32_t u = 0; u < W; ++u) {
            size_t i = (size_t)v * W + u;
            Rng g(seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(i + 1)));
            P16& p = out[i];
            if ((uint32_t)(g.uni() * 1000.0) < nan_permille) { p.x = p.y = p.z = qnan; p.rgba = 0; continue; }
            double d[3] = {((double)u - cx) / f, ((double)v - cy) / f, 1.0};
            double best = 1e30;
            const uint8_t* col = wall_back;
            // planes: back wall z = 3, floor y = +1, left wall x = -1.5
            { double t = 3.0; if (t < best) { best = t; col = wall_back; } }
            if (d[1] > 1e-9) { double t = 1.0 / d[1]; if (t < best) { best = t; col = floor_c; } }
            if (d[0] < -1e-9) { double t = -1.5 / d[0]; if (t < best) { best = t; col = wall_left; } }
            for (const Box& b : boxes) {   // slab test, ray origin 0
                double tn = 0.0, tf = 1e30;
                bool hit = true;
                for (int a = 0; a < 3 && hit; ++a) {
                    if (d[a] > -1e-12 && d[a] < 1e-12) { if (b.lo[a] > 0 || b.hi[a] < 0) hit = false; continue; }
                    double t1 = b.lo[a] / d[a], t2 = b.hi[a] / d[a];
                    if (t1 > t2) std::swap(t1, t2);
                    if (t1 > tn) tn = t1;
                    if (t2 < tf) tf = t2;
                    if (tn > tf) hit = false;
                }
                if (hit && tn > 1e-6 && tn < best) { best = tn; col = b.col; }
            }
            for (const Sphere& s : spheres) {
                double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                double dc = d[0] * s.c[0] + d[1] * s.c[1] + d[2] * s.c[2];
                double cc = s.c[0] * s.c[0] + s.c[1] * s.c[1] + s.c[2] * s.c[2] - s.r * s.r;
                double disc = dc * dc - dd * cc;
                if (disc <= 0) continue;
                double t = (dc - __builtin_sqrt(disc)) / dd;
                if (t > 1e-6 && t < best) { best = t; col = s.col; }
            }
            double z = best;   // d[2] == 1
            z += g.gauss() * 0.0012 * z * z;
            if (z < 0.3) z = 0.3;
            p.x = (float)(d[0] * z); p.y = (float)(d[1] * z); p.z = (float)z;
            uint8_t r = jitter_col(col[0], g, 8), gg = jitter_col(col[1], g, 8), b = jitter_col(col[2], g, 8);
            p.rgba = ((uint32_t)r << 16) | ((uint32_t)gg << 8) | (uint32_t)b;
        }
}

struct Face { double o[3], e1[3], e2[3]; double area; uint8_t ca[3], cb[3]; };

void synth_fused(uint64_t seed, size_t n, P16* out) {
    Rng sg(seed * 0x2545F4914F6CDD1DULL + 29);
    std::vector<Face> faces;
    auto add_box = [&](const double lo[3], const double hi[3], const uint8_t ca[3], const uint8_t cb[3]) {
        for (int a = 0; a < 3; ++a)
            for (int side = 0; side < 2; ++side) {
                Face fc;
                int b = (a + 1) % 3, c = (a + 2) % 3;
                for (int k = 0; k < 3; ++k) { fc.o[k] = lo[k]; fc.e1[k] = 0; fc.e2[k] = 0; }
                fc.o[a] = side ? hi[a] : lo[a];
                fc.e1[b] = hi[b] - lo[b];
                fc.e2[c] = hi[c] - lo[c];
                fc.area = fc.e1[b] * fc.e2[c];
                for (int k = 0; k < 3; ++k) { fc.ca[k] = ca[k]; fc.cb[k] = cb[k]; }
                faces.push_back(fc);
            }
    };
    const double z0 = 0.5;   // translate so every z > 0 (main() would fold negative z, :317-321)
    { double lo[3] = {-4, -1.5, z0}, hi[3] = {4, 1.5, z0 + 6}; uint8_t ca[3] = {210, 205, 190}, cb[3] = {180, 175, 165}; add_box(lo, hi, ca, cb); }
    for (int i = 0; i < 40; ++i) {
        double c[3] = {sg.uni(-3.5, 3.5), 0, sg.uni(z0 + 0.5, z0 + 5.5)};
        double hs[3] = {sg.uni(0.15, 0.6), sg.uni(0.15, 0.7), sg.uni(0.15, 0.6)};
        double lo[3], hi[3];
        for (int a = 0; a < 3; ++a) { lo[a] = c[a] - hs[a]; hi[a] = c[a] + hs[a]; }
        lo[1] = 1.5 - 2 * hs[1]; hi[1] = 1.5;   // standing on the floor (y down)
        uint8_t ca[3], cb[3];
        for (int a = 0; a < 3; ++a) { ca[a] = (uint8_t)(30 + (int)(sg.uni() * 200)); cb[a] = (uint8_t)(30 + (int)(sg.uni() * 200)); }
        add_box(lo, hi, ca, cb);
    }
    std::vector<double> cum(faces.size());
    double tot = 0;
    for (size_t i = 0; i < faces.size(); ++i) { tot += faces[i].area; cum[i] = tot; }
    for (size_t i = 0; i < n; ++i) {
        Rng g(seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(i + 1)));
        double pick = g.uni() * tot;
        size_t fi = (size_t)(std::lower_bound(cum.begin(), cum.end(), pick) - cum.begin());
        if (fi >= faces.size()) fi = faces.size() - 1;
        const Face& fc = faces[fi];
        double a = g.uni(), b = g.uni();
        double p[3];
        double l1 = 0, l2 = 0;
        for (int k = 0; k < 3; ++k) { p[k] = fc.o[k] + a * fc.e1[k] + b * fc.e2[k]; l1 += fc.e1[k]; l2 += fc.e2[k]; }
        for (int k = 0; k < 3; ++k) p[k] += g.gauss() * 0.002;
        if (p[2] < 0.05) p[2] = 0.05;
        int chk = ((int)(a * l1 / 0.5) + (int)(b * l2 / 0.5)) & 1;
        const uint8_t* col = chk ? fc.ca : fc.cb;
        out[i].x = (float)p[0]; out[i].y = (float)p[1]; out[i].z = (float)p[2];
        uint8_t r = jitter_col(col[0], g, 6), gg = jitter_col(col[1], g, 6), bb = jitter_col(col[2], g, 6);
        out[i].rgba = ((uint32_t)r << 16) | ((uint32_t)gg << 8) | (uint32_t)bb;
    }
}

}  // namespace

extern "C" int f3ds_synth_frame(int kind, uint64_t seed, uint32_t width, uint32_t height, uint32_t nan_permille,
                                void* points16) {
    if (!points16 || width == 0 || height == 0) return F3DS_ERR_ARG;
    if (kind == 0) synth_rgbd(seed, width, height, nan_permille, (P16*)points16);
    else if (kind == 1) synth_fused(seed, (size_t)width * height, (P16*)points16);
    else return F3DS_ERR_ARG;
    return F3DS_OK;
}
