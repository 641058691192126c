// Native .npz ingest / output for many samples.
//
// The reference reads one converted sample after the other with np.load inside the driver loop
// (wisecondor.py:75-80, 193-196) and writes one result file per `test` process with
// np.savez_compressed (wisecondor.py:270-280).  Once the GPU side takes microseconds per sample,
// the zip inflate + unpickle of the chromosome dict and the deflate of the result arrays are the
// ceiling, and in Python both hold the GIL.  Here both run in plain C++ threads:
//
//   wc_read_samples        N sample files -> dense int32 count rows (pad / truncate to the
//                          reference's chromosome lengths, toNumpyRefFormat's first half,
//                          wisetools.py:268-274; scaleSample's bin merging, wisetools.py:220-237)
//   wc_write_test_results  N result rows -> N `test` output files with the keys, dtypes and shapes of
//                          the reference's: a zip of .npy members, the ragged
//                          per-chromosome arrays as pickled object arrays
//
// The reader carries a small pickle machine (the opcodes numpy emits for a dict of integer arrays
// under protocols 2 - 4, Python 2 and 3 writers); anything it does not know makes that FILE
// report WC_NPZ_UNSUPPORTED and the Python caller reads it with np.load instead -- never a guess.
// No GPU code in this file.
#include <zlib.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wisecondor_hip.h"

namespace {

// ------------------------------------------------------------------ zip ----
struct ZipEntry {
    std::string name;
    int method = 0;
    uint32_t crc = 0;
    uint64_t csize = 0, usize = 0, local = 0;
};

inline uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const unsigned char *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

bool read_file(const char *path, std::string &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return false; }
    long n = ftell(f);
    if (n < 0) { fclose(f); return false; }
    rewind(f);
    out.resize((size_t)n);
    size_t got = n ? fread(&out[0], 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

// Largest member this reader inflates: a converted sample is a few MB; anything beyond this is a damaged header
// (the file is then left to np.load, which words the error).
constexpr uint64_t MAX_MEMBER_BYTES = 1ull << 30;

bool zip_directory(const std::string &buf, std::vector<ZipEntry> &out) {
    const unsigned char *b = (const unsigned char *)buf.data();
    const size_t n = buf.size();
    if (n < 22) return false;
    size_t eocd = n;
    for (size_t back = 22; back <= n && back <= 22 + 65535; ++back)
        if (rd32(b + n - back) == 0x06054b50u) { eocd = n - back; break; }
    if (eocd == n) return false;
    uint64_t count = rd16(b + eocd + 10), cd_off = rd32(b + eocd + 16);
    if ((count == 0xFFFF || cd_off == 0xFFFFFFFFu) && eocd >= 20 && rd32(b + eocd - 20) == 0x07064b50u) {
        const uint64_t z64 = rd64(b + eocd - 20 + 8);        // zip64 end of central directory record
        // (every offset and size below comes from the file: compare without additions that could wrap)
        if (n < 56 || z64 > n - 56 || rd32(b + z64) != 0x06064b50u) return false;
        count = rd64(b + z64 + 32);
        cd_off = rd64(b + z64 + 48);
    }
    if (cd_off > n || count > n / 46) return false;         // a directory entry takes at least 46 bytes
    size_t at = (size_t)cd_off;
    for (uint64_t e = 0; e < count; ++e) {
        if (n - at < 46 || rd32(b + at) != 0x02014b50u) return false;
        ZipEntry z;
        z.method = rd16(b + at + 10);
        z.crc = rd32(b + at + 16);
        z.csize = rd32(b + at + 20);
        z.usize = rd32(b + at + 24);
        const size_t nl = rd16(b + at + 28), xl = rd16(b + at + 30), cl = rd16(b + at + 32);
        z.local = rd32(b + at + 42);
        if (n - at - 46 < nl + xl + cl) return false;
        z.name.assign((const char *)b + at + 46, nl);
        // zip64 extra field: the values that overflowed, in the order usize, csize, local offset
        size_t x = at + 46 + nl;
        const size_t xe = x + xl;
        while (x + 4 <= xe) {
            const uint16_t id = rd16(b + x), len = rd16(b + x + 2);
            if (id == 0x0001) {
                size_t q = x + 4;
                if (z.usize == 0xFFFFFFFFu && q + 8 <= xe) { z.usize = rd64(b + q); q += 8; }
                if (z.csize == 0xFFFFFFFFu && q + 8 <= xe) { z.csize = rd64(b + q); q += 8; }
                if (z.local == 0xFFFFFFFFu && q + 8 <= xe) { z.local = rd64(b + q); q += 8; }
            }
            x += 4 + len;
        }
        // a member cannot be larger than the file that holds it (stored) / a sane multiple of it (deflated), and its
        // header must lie inside the file
        if (z.csize > n || z.local > n || z.usize > MAX_MEMBER_BYTES) return false;
        out.push_back(z);
        at += 46 + nl + xl + cl;
    }
    return true;
}

bool zip_member(const std::string &buf, const ZipEntry &z, std::string &out) {
    const unsigned char *b = (const unsigned char *)buf.data();
    const size_t n = buf.size();
    if (n < 30 || z.local > n - 30 || rd32(b + z.local) != 0x04034b50u) return false;
    const size_t data = (size_t)z.local + 30 + rd16(b + z.local + 26) + rd16(b + z.local + 28);   // <= n + 2 * 65535: no wrap
    if (data > n || z.csize > n - data) return false;
    if (z.usize > MAX_MEMBER_BYTES || z.csize > 0xFFFFFFFFull || z.usize > 0xFFFFFFFFull) return false;   // zlib's 32-bit counters
    out.resize((size_t)z.usize);
    if (z.method == 0) {
        if (z.csize != z.usize) return false;
        memcpy(&out[0], b + data, (size_t)z.usize);
        return true;
    }
    if (z.method != 8) return false;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(b + data);
    zs.avail_in = (uInt)z.csize;
    zs.next_out = (Bytef *)&out[0];
    zs.avail_out = (uInt)z.usize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.total_out == z.usize;
}

// --------------------------------------------------------------- pickle ----
struct Val;
typedef std::shared_ptr<Val> VP;
struct Val {
    enum Kind { NONE, BOOL, INT, FLOAT, BYTES, STR, TUPLE, LIST, DICT, GLOBAL, NDARRAY, DTYPE, OTHER, MARK } kind = NONE;
    int64_t i = 0;
    double f = 0.0;
    std::string s;              // BYTES / STR / GLOBAL "module name" / DTYPE code ("i4") / NDARRAY raw data
    std::vector<VP> items;      // TUPLE / LIST; DICT: key, value, key, value ...; NDARRAY of objects: the elements
    std::vector<int64_t> shape; // NDARRAY
    std::string dtype;          // NDARRAY: code of its dtype
    char endian = '<';          // DTYPE / NDARRAY
};
VP mk(Val::Kind k) { VP v = std::make_shared<Val>(); v->kind = k; return v; }

struct Unpickler {
    const unsigned char *p, *end;
    std::vector<VP> stack, memo;
    bool ok = true;
    explicit Unpickler(const std::string &s) : p((const unsigned char *)s.data()), end(p + s.size()) {}
    bool need(size_t n) { if ((size_t)(end - p) < n) ok = false; return ok; }
    VP pop() { if (stack.empty()) { ok = false; return mk(Val::NONE); } VP v = stack.back(); stack.pop_back(); return v; }
    std::vector<VP> pop_mark() {
        std::vector<VP> out;
        size_t at = stack.size();
        while (at > 0 && stack[at - 1]->kind != Val::MARK) --at;
        if (at == 0) { ok = false; return out; }
        out.assign(stack.begin() + at, stack.end());
        stack.resize(at - 1);
        return out;
    }
    void memo_put(size_t idx, VP v) { if (memo.size() <= idx) memo.resize(idx + 1); memo[idx] = v; }
    VP str_of(size_t n, Val::Kind k) { VP v = mk(k); if (need(n)) { v->s.assign((const char *)p, n); p += n; } return v; }
    std::string line() { std::string s; while (p < end && *p != '\n') s.push_back((char)*p++); if (p < end) ++p; else ok = false; return s; }

    VP reduce(const VP &fn, const VP &args) {
        if (fn->kind != Val::GLOBAL || args->kind != Val::TUPLE) return mk(Val::OTHER);
        const std::string &g = fn->s;
        const bool np_core = g.rfind("numpy.core.multiarray ", 0) == 0 || g.rfind("numpy._core.multiarray ", 0) == 0;
        if (np_core && g.size() >= 12 && g.compare(g.size() - 12, 12, "_reconstruct") == 0) return mk(Val::NDARRAY);
        if (g == "numpy dtype" && !args->items.empty() && (args->items[0]->kind == Val::STR || args->items[0]->kind == Val::BYTES)) {
            VP d = mk(Val::DTYPE);
            d->s = args->items[0]->s;
            return d;
        }
        if (g == "_codecs encode" && !args->items.empty() && args->items[0]->kind == Val::STR) {
            // numpy under Python 3 with protocol 2: bytes travel as a latin1-decoded unicode string
            VP b = mk(Val::BYTES);
            const std::string &u = args->items[0]->s;
            for (size_t k = 0; k < u.size(); ++k) {
                unsigned char c = (unsigned char)u[k];
                if (c < 0x80) b->s.push_back((char)c);
                else if ((c & 0xE0) == 0xC0 && k + 1 < u.size()) { b->s.push_back((char)(((c & 0x1F) << 6) | ((unsigned char)u[k + 1] & 0x3F))); ++k; }
                else { ok = false; break; }
            }
            return b;
        }
        if (np_core && g.size() >= 6 && g.compare(g.size() - 6, 6, "scalar") == 0 && args->items.size() == 2 &&
            args->items[0]->kind == Val::DTYPE && (args->items[1]->kind == Val::BYTES || args->items[1]->kind == Val::STR)) {
            const std::string &code = args->items[0]->s, &raw = args->items[1]->s;
            if (code == "f8" && raw.size() == 8) { VP v = mk(Val::FLOAT); memcpy(&v->f, raw.data(), 8); return v; }
            if (code == "f4" && raw.size() == 4) { VP v = mk(Val::FLOAT); float t; memcpy(&t, raw.data(), 4); v->f = t; return v; }
            if (code == "i8" && raw.size() == 8) { VP v = mk(Val::INT); memcpy(&v->i, raw.data(), 8); return v; }
            if (code == "i4" && raw.size() == 4) { VP v = mk(Val::INT); int32_t t; memcpy(&t, raw.data(), 4); v->i = t; return v; }
        }
        return mk(Val::OTHER);
    }
    void build(const VP &obj, const VP &state) {
        if (obj->kind == Val::NDARRAY && state->kind == Val::TUPLE && state->items.size() >= 5) {
            const VP &shape = state->items[1], &dt = state->items[2], &raw = state->items[4];
            if (shape->kind != Val::TUPLE || dt->kind != Val::DTYPE) { ok = false; return; }
            for (const VP &d : shape->items) { if (d->kind != Val::INT) { ok = false; return; } obj->shape.push_back(d->i); }
            obj->dtype = dt->s;
            obj->endian = dt->endian;
            obj->i = state->items[3]->i;      // fortran flag
            if (raw->kind == Val::BYTES || raw->kind == Val::STR) obj->s = raw->s;
            else if (raw->kind == Val::LIST) obj->items = raw->items;
            else ok = false;
        } else if (obj->kind == Val::DTYPE && state->kind == Val::TUPLE && state->items.size() >= 2 &&
                   (state->items[1]->kind == Val::STR || state->items[1]->kind == Val::BYTES) && !state->items[1]->s.empty()) {
            obj->endian = state->items[1]->s[0];
        }
    }
    VP run() {
        while (ok && p < end) {
            const unsigned char op = *p++;
            switch (op) {
                case 0x80: need(1); p += 1; break;                               // PROTO
                case 0x95: need(8); p += 8; break;                               // FRAME
                case '.': return stack.empty() ? VP() : stack.back();            // STOP
                case 'N': stack.push_back(mk(Val::NONE)); break;
                case 0x88: { VP v = mk(Val::BOOL); v->i = 1; stack.push_back(v); break; }
                case 0x89: { VP v = mk(Val::BOOL); v->i = 0; stack.push_back(v); break; }
                case 'J': { VP v = mk(Val::INT); if (need(4)) { v->i = (int32_t)rd32(p); p += 4; } stack.push_back(v); break; }
                case 'K': { VP v = mk(Val::INT); if (need(1)) { v->i = *p; p += 1; } stack.push_back(v); break; }
                case 'M': { VP v = mk(Val::INT); if (need(2)) { v->i = rd16(p); p += 2; } stack.push_back(v); break; }
                case 0x8a: {                                                      // LONG1
                    VP v = mk(Val::INT);
                    if (need(1)) {
                        const size_t n = *p++;
                        if (n > 8 || !need(n)) { ok = false; break; }
                        uint64_t u = 0;
                        for (size_t k = 0; k < n; ++k) u |= (uint64_t)p[k] << (8 * k);
                        if (n && n < 8 && (p[n - 1] & 0x80)) u |= ~0ull << (8 * n);
                        v->i = (int64_t)u;
                        p += n;
                    }
                    stack.push_back(v);
                    break;
                }
                case 'I': case 'L': {                                              // INT / LONG (text)
                    std::string t = line();
                    VP v = mk(Val::INT);
                    if (t == "01") { v->kind = Val::BOOL; v->i = 1; }
                    else if (t == "00") { v->kind = Val::BOOL; v->i = 0; }
                    else v->i = strtoll(t.c_str(), nullptr, 10);
                    stack.push_back(v);
                    break;
                }
                case 'G': {                                                        // BINFLOAT, big endian
                    VP v = mk(Val::FLOAT);
                    if (need(8)) { unsigned char t[8]; for (int k = 0; k < 8; ++k) t[k] = p[7 - k]; memcpy(&v->f, t, 8); p += 8; }
                    stack.push_back(v);
                    break;
                }
                case 'F': { VP v = mk(Val::FLOAT); v->f = strtod(line().c_str(), nullptr); stack.push_back(v); break; }
                case 'U': { size_t n = 0; if (need(1)) n = *p++; stack.push_back(str_of(n, Val::STR)); break; }          // SHORT_BINSTRING
                case 'T': { size_t n = 0; if (need(4)) { n = rd32(p); p += 4; } stack.push_back(str_of(n, Val::STR)); break; }   // BINSTRING
                case 'X': { size_t n = 0; if (need(4)) { n = rd32(p); p += 4; } stack.push_back(str_of(n, Val::STR)); break; }   // BINUNICODE
                case 0x8c: { size_t n = 0; if (need(1)) n = *p++; stack.push_back(str_of(n, Val::STR)); break; }
                case 0x8d: { size_t n = 0; if (need(8)) { n = (size_t)rd64(p); p += 8; } stack.push_back(str_of(n, Val::STR)); break; }
                case 'B': { size_t n = 0; if (need(4)) { n = rd32(p); p += 4; } stack.push_back(str_of(n, Val::BYTES)); break; }
                case 'C': { size_t n = 0; if (need(1)) n = *p++; stack.push_back(str_of(n, Val::BYTES)); break; }
                case 0x8e: { size_t n = 0; if (need(8)) { n = (size_t)rd64(p); p += 8; } stack.push_back(str_of(n, Val::BYTES)); break; }
                case ')': stack.push_back(mk(Val::TUPLE)); break;
                case 't': { VP v = mk(Val::TUPLE); v->items = pop_mark(); stack.push_back(v); break; }
                case 0x85: { VP v = mk(Val::TUPLE); VP a = pop(); v->items = {a}; stack.push_back(v); break; }
                case 0x86: { VP v = mk(Val::TUPLE); VP b = pop(), a = pop(); v->items = {a, b}; stack.push_back(v); break; }
                case 0x87: { VP v = mk(Val::TUPLE); VP c = pop(), b = pop(), a = pop(); v->items = {a, b, c}; stack.push_back(v); break; }
                case ']': stack.push_back(mk(Val::LIST)); break;
                case 'l': { VP v = mk(Val::LIST); v->items = pop_mark(); stack.push_back(v); break; }
                case 'a': { VP x = pop(); if (stack.empty() || stack.back()->kind != Val::LIST) { ok = false; break; } stack.back()->items.push_back(x); break; }
                case 'e': { std::vector<VP> xs = pop_mark(); if (stack.empty() || stack.back()->kind != Val::LIST) { ok = false; break; }
                            for (VP &x : xs) stack.back()->items.push_back(x); break; }
                case '}': stack.push_back(mk(Val::DICT)); break;
                case 'd': { VP v = mk(Val::DICT); v->items = pop_mark(); stack.push_back(v); break; }
                case 's': { VP val = pop(), key = pop(); if (stack.empty() || stack.back()->kind != Val::DICT) { ok = false; break; }
                            stack.back()->items.push_back(key); stack.back()->items.push_back(val); break; }
                case 'u': { std::vector<VP> xs = pop_mark(); if (stack.empty() || stack.back()->kind != Val::DICT || (xs.size() & 1)) { ok = false; break; }
                            for (VP &x : xs) stack.back()->items.push_back(x); break; }
                case '(': stack.push_back(mk(Val::MARK)); break;
                case 'c': { VP v = mk(Val::GLOBAL); std::string m = line(), n = line(); v->s = m + " " + n; stack.push_back(v); break; }
                case 0x93: { VP n = pop(), m = pop(); VP v = mk(Val::GLOBAL); v->s = m->s + " " + n->s; stack.push_back(v); break; }
                case 'R': { VP args = pop(), fn = pop(); stack.push_back(reduce(fn, args)); break; }
                case 0x81: { pop(); pop(); stack.push_back(mk(Val::OTHER)); break; }  // NEWOBJ
                case 'b': { VP state = pop(); if (stack.empty()) { ok = false; break; } build(stack.back(), state); break; }
                case 'q': { if (need(1) && !stack.empty()) memo_put(*p, stack.back()); p += 1; break; }
                case 'r': { if (need(4) && !stack.empty()) memo_put(rd32(p), stack.back()); p += 4; break; }
                case 0x94: { if (stack.empty()) { ok = false; break; } memo.push_back(stack.back()); break; }
                case 'h': { if (need(1)) { size_t k = *p++; if (k < memo.size() && memo[k]) stack.push_back(memo[k]); else ok = false; } break; }
                case 'j': { if (need(4)) { size_t k = rd32(p); p += 4; if (k < memo.size() && memo[k]) stack.push_back(memo[k]); else ok = false; } break; }
                case '0': pop(); break;
                case '2': if (stack.empty()) ok = false; else stack.push_back(stack.back()); break;
                default: ok = false; break;           // an opcode numpy's writers do not use for these files
            }
        }
        ok = false;     // ran off the end without STOP
        return VP();
    }
};

// the pickled object of a 0-d object array member -> the dict it holds
bool npy_object_dict(const std::string &npy, VP &dict) {
    if (npy.size() < 10 || memcmp(npy.data(), "\x93NUMPY", 6) != 0) return false;
    const unsigned char *b = (const unsigned char *)npy.data();
    size_t hl, off;
    if (b[6] == 1) { hl = rd16(b + 8); off = 10; } else { if (npy.size() < 12) return false; hl = rd32(b + 8); off = 12; }
    if (off + hl > npy.size()) return false;
    const std::string header(npy.data() + off, hl);
    if (header.find("'|O'") == std::string::npos && header.find("'O'") == std::string::npos) return false;
    const std::string body = npy.substr(off + hl);      // the unpickler keeps pointers into it
    Unpickler u(body);
    VP root = u.run();
    if (!root) return false;
    // np.save pickles the array itself: a 0-d object ndarray whose element list holds the dict
    if (root->kind == Val::NDARRAY && root->items.size() == 1) root = root->items[0];
    if (root->kind != Val::DICT) return false;
    dict = root;
    return true;
}

VP dict_get(const VP &d, const char *key) {
    for (size_t k = 0; k + 1 < d->items.size(); k += 2)
        if ((d->items[k]->kind == Val::STR || d->items[k]->kind == Val::BYTES) && d->items[k]->s == key) return d->items[k + 1];
    return VP();
}

// element e of a 1-d integer array as int64 (the dtypes `convert` and numpy's int defaults produce)
bool array_values(const VP &a, std::vector<int64_t> &out) {
    if (!a || a->kind != Val::NDARRAY || a->shape.size() != 1 || a->endian == '>') return false;
    const int64_t n = a->shape[0];
    const std::string &c = a->dtype;
    size_t w = 0;
    if (c == "i4" || c == "u4") w = 4; else if (c == "i8" || c == "u8") w = 8; else if (c == "i2" || c == "u2") w = 2;
    else if (c == "f8") w = 8; else if (c == "f4") w = 4; else return false;
    if (n < 0 || (uint64_t)n > a->s.size() / w || a->s.size() != (size_t)n * w) return false;   // (n * w cannot wrap after the first test)
    out.resize((size_t)n);
    const char *p = a->s.data();
    for (int64_t e = 0; e < n; ++e, p += w) {
        if (c == "i4") { int32_t t; memcpy(&t, p, 4); out[e] = t; }
        else if (c == "u4") { uint32_t t; memcpy(&t, p, 4); out[e] = t; }
        else if (c == "i8" || c == "u8") { int64_t t; memcpy(&t, p, 8); out[e] = t; }
        else if (c == "i2") { int16_t t; memcpy(&t, p, 2); out[e] = t; }
        else if (c == "u2") { uint16_t t; memcpy(&t, p, 2); out[e] = t; }
        else if (c == "f8") { double t; memcpy(&t, p, 8); out[e] = (int64_t)t; }
        else { float t; memcpy(&t, p, 4); out[e] = (int64_t)t; }
    }
    return true;
}

int read_one(const char *path, const int64_t *chrom_sizes, int n_chrom, double to_binsize, int32_t *row,
             double *binsize_out) {
    std::string buf;
    if (!read_file(path, buf)) return WC_NPZ_IO;
    std::vector<ZipEntry> dir;
    if (!zip_directory(buf, dir)) return WC_NPZ_UNSUPPORTED;
    const ZipEntry *zs = nullptr, *za = nullptr;
    for (const ZipEntry &z : dir) {
        if (z.name == "sample.npy") zs = &z;
        if (z.name == "arguments.npy") za = &z;
    }
    if (!zs || !za) return WC_NPZ_UNSUPPORTED;
    std::string m;
    VP args, sample;
    if (!zip_member(buf, *za, m) || !npy_object_dict(m, args)) return WC_NPZ_UNSUPPORTED;
    if (!zip_member(buf, *zs, m) || !npy_object_dict(m, sample)) return WC_NPZ_UNSUPPORTED;
    VP bs = dict_get(args, "binsize");
    if (!bs || (bs->kind != Val::FLOAT && bs->kind != Val::INT)) return WC_NPZ_UNSUPPORTED;
    const double own = bs->kind == Val::FLOAT ? bs->f : (double)bs->i;
    *binsize_out = own;
    int64_t scale = 1;
    if (to_binsize > 0 && own != to_binsize) {
        // scaleSample (wisetools.py:220-237): whole multiples only; the caller reports anything else
        if (own <= 0 || to_binsize < own || std::fmod(to_binsize, own) != 0.0) return WC_NPZ_BINSIZE;
        scale = (int64_t)(to_binsize / own);
    }
    int64_t at = 0;
    std::vector<int64_t> vals;
    for (int c = 0; c < n_chrom; ++c) {
        char key[16];
        snprintf(key, sizeof(key), "%d", c + 1);
        if (!array_values(dict_get(sample, key), vals)) return WC_NPZ_UNSUPPORTED;
        const int64_t want = chrom_sizes[c];
        const int64_t have_bins = ((int64_t)vals.size() + scale - 1) / scale;
        const int64_t have = have_bins < want ? have_bins : want;
        for (int64_t b = 0; b < have; ++b) {
            int64_t sum = 0;
            const int64_t lo = b * scale, hi = lo + scale < (int64_t)vals.size() ? lo + scale : (int64_t)vals.size();
            for (int64_t e = lo; e < hi; ++e) sum += vals[(size_t)e];
            row[at + b] = (int32_t)sum;
        }
        for (int64_t b = have; b < want; ++b) row[at + b] = 0;
        at += want;
    }
    return WC_OK;
}

// chromosome lengths of one sample file (what toNumpyArray needs before it can size the dense matrix:
// the reference takes the per-chromosome maximum over the samples, wisetools.py:243-250)
int lengths_one(const char *path, int n_chrom, double to_binsize, int64_t *len_out, double *binsize_out) {
    std::string buf;
    if (!read_file(path, buf)) return WC_NPZ_IO;
    std::vector<ZipEntry> dir;
    if (!zip_directory(buf, dir)) return WC_NPZ_UNSUPPORTED;
    const ZipEntry *zs = nullptr, *za = nullptr;
    for (const ZipEntry &z : dir) {
        if (z.name == "sample.npy") zs = &z;
        if (z.name == "arguments.npy") za = &z;
    }
    if (!zs || !za) return WC_NPZ_UNSUPPORTED;
    std::string m;
    VP args, sample;
    if (!zip_member(buf, *za, m) || !npy_object_dict(m, args)) return WC_NPZ_UNSUPPORTED;
    if (!zip_member(buf, *zs, m) || !npy_object_dict(m, sample)) return WC_NPZ_UNSUPPORTED;
    VP bs = dict_get(args, "binsize");
    if (!bs || (bs->kind != Val::FLOAT && bs->kind != Val::INT)) return WC_NPZ_UNSUPPORTED;
    const double own = bs->kind == Val::FLOAT ? bs->f : (double)bs->i;
    *binsize_out = own;
    int64_t scale = 1;
    if (to_binsize > 0 && own != to_binsize) {
        if (own <= 0 || to_binsize < own || std::fmod(to_binsize, own) != 0.0) return WC_NPZ_BINSIZE;
        scale = (int64_t)(to_binsize / own);
    }
    for (int c = 0; c < n_chrom; ++c) {
        char key[16];
        snprintf(key, sizeof(key), "%d", c + 1);
        VP a = dict_get(sample, key);
        if (!a || a->kind != Val::NDARRAY || a->shape.size() != 1) return WC_NPZ_UNSUPPORTED;
        len_out[c] = (a->shape[0] + scale - 1) / scale;       // scaleSample: ceil(len / scale) merged bins
    }
    return WC_OK;
}

// --------------------------------------------------------------- writer ----
void put16(std::string &s, uint16_t v) { s.push_back((char)(v & 0xFF)); s.push_back((char)(v >> 8)); }
void put32(std::string &s, uint32_t v) { for (int k = 0; k < 4; ++k) s.push_back((char)((v >> (8 * k)) & 0xFF)); }

std::string npy_header(const std::string &descr, const std::string &shape) {
    std::string dict = "{'descr': '" + descr + "', 'fortran_order': False, 'shape': " + shape + ", }";
    // magic (6) + version (2) + header length (2) + dict + padding + newline: a multiple of 64
    size_t total = 10 + dict.size() + 1;
    const size_t pad = (64 - total % 64) % 64;
    dict.append(pad, ' ');
    dict.push_back('\n');
    std::string out("\x93NUMPY\x01\x00", 8);
    put16(out, (uint16_t)dict.size());
    return out + dict;
}

std::string npy_f64(const double *v, const std::vector<int64_t> &shape) {
    std::string sh = "(";
    int64_t n = 1;
    for (size_t k = 0; k < shape.size(); ++k) {
        sh += std::to_string(shape[k]);
        if (k + 1 < shape.size() || shape.size() == 1) sh += (shape.size() == 1 ? "," : ", ");
        n *= shape[k];
    }
    sh += ")";
    std::string out = npy_header("<f8", sh);
    out.append((const char *)v, (size_t)n * 8);
    return out;
}

// protocol 3 pickle of a 1-d object ndarray holding float64 arrays (what np.save writes for the
// ragged per-chromosome lists); opcodes only, no memo
void pk_global(std::string &s, const char *mod, const char *name) { s.push_back('c'); s += mod; s.push_back('\n'); s += name; s.push_back('\n'); }
void pk_int(std::string &s, int64_t v) {
    if (v >= 0 && v < 256) { s.push_back('K'); s.push_back((char)v); }
    else if (v >= 0 && v < 65536) { s.push_back('M'); put16(s, (uint16_t)v); }
    else { s.push_back('J'); put32(s, (uint32_t)(int32_t)v); }
}
void pk_unicode(std::string &s, const char *t) { s.push_back('X'); put32(s, (uint32_t)strlen(t)); s += t; }
void pk_dtype(std::string &s, const char *code, const char *endian, int flags) {
    pk_global(s, "numpy", "dtype");
    pk_unicode(s, code); s.push_back((char)0x89); s.push_back((char)0x88); s.push_back((char)0x87); s.push_back('R');   // (code, False, True) REDUCE
    s.push_back('(');
    pk_int(s, 3); pk_unicode(s, endian); s.push_back('N'); s.push_back('N'); s.push_back('N');
    s.push_back('J'); put32(s, 0xFFFFFFFFu); s.push_back('J'); put32(s, 0xFFFFFFFFu); pk_int(s, flags);
    s.push_back('t'); s.push_back('b');
}
void pk_reconstruct(std::string &s) {
    pk_global(s, "numpy.core.multiarray", "_reconstruct");
    pk_global(s, "numpy", "ndarray");
    pk_int(s, 0); s.push_back((char)0x85);                      // (0,)
    s.push_back('C'); s.push_back(1); s.push_back('b');         // b'b'
    s.push_back((char)0x87); s.push_back('R');
}
std::string npy_object_of_f64(const double *row, const int64_t *sizes, int n) {
    std::string p;
    p.push_back((char)0x80); p.push_back(3);
    pk_reconstruct(p);
    p.push_back('(');
    pk_int(p, 1);
    pk_int(p, n); p.push_back((char)0x85);
    pk_dtype(p, "O8", "|", 63);
    p.push_back((char)0x89);
    p.push_back(']');
    p.push_back('(');
    int64_t at = 0;
    for (int c = 0; c < n; ++c) {
        pk_reconstruct(p);
        p.push_back('(');
        pk_int(p, 1);
        pk_int(p, sizes[c]); p.push_back((char)0x85);
        pk_dtype(p, "f8", "<", 0);
        p.push_back((char)0x89);
        p.push_back('B'); put32(p, (uint32_t)(sizes[c] * 8)); p.append((const char *)(row + at), (size_t)sizes[c] * 8);
        p.push_back('t'); p.push_back('b');
        at += sizes[c];
    }
    p.push_back('e');
    p.push_back('t'); p.push_back('b');
    p.push_back('.');
    return npy_header("|O", "(" + std::to_string(n) + ",)") + p;
}

struct ZipWriter {
    std::string out, central;
    int count = 0;
    int level;
    explicit ZipWriter(int lvl) : level(lvl) {}
    bool add(const std::string &name, const std::string &data, bool floats = false) {
        const uint32_t crc = (uint32_t)crc32(0L, (const Bytef *)data.data(), (uInt)data.size());
        std::string comp;
        int method = 0;
        if (level > 0 && data.size() > 64) {
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            // level 1: run-length strategy -- on float64 noise with runs of exact zeros (masked bins) it
            // reaches the default strategy's ratio (85 %) at three times its speed (string matching finds
            // nothing else in such data); higher levels: zlib's default strategy, what np.savez_compressed uses
            if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, (level == 1 && floats) ? Z_RLE : Z_DEFAULT_STRATEGY) != Z_OK) return false;
            comp.resize(deflateBound(&zs, (uLong)data.size()));
            zs.next_in = (Bytef *)data.data();
            zs.avail_in = (uInt)data.size();
            zs.next_out = (Bytef *)&comp[0];
            zs.avail_out = (uInt)comp.size();
            const int rc = deflate(&zs, Z_FINISH);
            comp.resize(zs.total_out);
            deflateEnd(&zs);
            if (rc != Z_STREAM_END) return false;
            method = 8;
        }
        const std::string &body = method ? comp : data;
        const uint32_t offset = (uint32_t)out.size();
        put32(out, 0x04034b50u); put16(out, 20); put16(out, 0); put16(out, (uint16_t)method);
        put16(out, 0); put16(out, 0x21);               // time 00:00:00, date 1980-01-01
        put32(out, crc); put32(out, (uint32_t)body.size()); put32(out, (uint32_t)data.size());
        put16(out, (uint16_t)name.size()); put16(out, 0);
        out += name;
        out += body;
        put32(central, 0x02014b50u); put16(central, 20); put16(central, 20); put16(central, 0); put16(central, (uint16_t)method);
        put16(central, 0); put16(central, 0x21);
        put32(central, crc); put32(central, (uint32_t)body.size()); put32(central, (uint32_t)data.size());
        put16(central, (uint16_t)name.size()); put16(central, 0); put16(central, 0); put16(central, 0); put16(central, 0);
        put32(central, 0x01800000u);                    // external attributes: a regular file, rw-------
        put32(central, offset);
        central += name;
        ++count;
        return true;
    }
    bool finish(const char *path) {
        const uint32_t cd_off = (uint32_t)out.size(), cd_len = (uint32_t)central.size();
        out += central;
        put32(out, 0x06054b50u); put16(out, 0); put16(out, 0); put16(out, (uint16_t)count); put16(out, (uint16_t)count);
        put32(out, cd_len); put32(out, cd_off); put16(out, 0);
        FILE *f = fopen(path, "wb");
        if (!f) return false;
        const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
        return fclose(f) == 0 && ok;
    }
};

template <class F>
void run_pool(int n, int threads, F work) {
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    std::atomic<int> next(0);
    auto loop = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) work(i); };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(loop);
    loop();
    for (std::thread &t : pool) t.join();
}

}  // namespace

extern "C" {

int wc_read_samples(const char *const *paths, int n_files, int n_threads, const int64_t *chrom_sizes, int n_chrom,
                    double to_binsize, int32_t *counts_out, int64_t row_stride, double *binsize_out, int *status) {
    if (!paths || n_files < 0 || !chrom_sizes || n_chrom <= 0 || !counts_out || !binsize_out || !status) return WC_E_ARG;
    int64_t total = 0;
    for (int c = 0; c < n_chrom; ++c) { if (chrom_sizes[c] < 0) return WC_E_ARG; total += chrom_sizes[c]; }
    if (row_stride < total) return WC_E_ARG;
    run_pool(n_files, n_threads, [&](int i) {
        binsize_out[i] = 0.0;
        int rc;
        try {
            rc = read_one(paths[i], chrom_sizes, n_chrom, to_binsize, counts_out + (int64_t)i * row_stride, &binsize_out[i]);
        } catch (...) {
            rc = WC_NPZ_UNSUPPORTED;
        }
        status[i] = rc;
    });
    return WC_OK;
}

int wc_read_sample_lengths(const char *const *paths, int n_files, int n_threads, int n_chrom, double to_binsize,
                           int64_t *lengths_out, double *binsize_out, int *status) {
    if (!paths || n_files < 0 || n_chrom <= 0 || !lengths_out || !binsize_out || !status) return WC_E_ARG;
    run_pool(n_files, n_threads, [&](int i) {
        binsize_out[i] = 0.0;
        int rc;
        try {
            rc = lengths_one(paths[i], n_chrom, to_binsize, lengths_out + (int64_t)i * n_chrom, &binsize_out[i]);
        } catch (...) {
            rc = WC_NPZ_UNSUPPORTED;
        }
        status[i] = rc;
    });
    return WC_OK;
}

int wc_write_test_results(int n_files, int n_threads, const char *const *out_paths, const unsigned char *const *args_npy,
                          const int64_t *args_len, const unsigned char *runtime_npy, int64_t runtime_len, double binsize,
                          int binsize_is_int,
                          double threshold_z, const int64_t *chrom_sizes, int n_chrom, const double *z, const double *r,
                          int64_t row_stride, const double *cwz, int n_sel, const double *calls, const int32_t *n_calls,
                          int max_calls, const double *asdef, int level, int *status) {
    if (!out_paths || n_files < 0 || !args_npy || !args_len || !runtime_npy || !chrom_sizes || n_chrom <= 0 || !z || !r ||
        !cwz || !calls || !n_calls || !asdef || !status)
        return WC_E_ARG;
    run_pool(n_files, n_threads, [&](int i) {
        int rc = WC_OK;
        try {
            ZipWriter zw(level);
            const std::vector<int64_t> none;
            bool ok = zw.add("arguments.npy", std::string((const char *)args_npy[i], (size_t)args_len[i]));
            ok = ok && zw.add("runtime.npy", std::string((const char *)runtime_npy, (size_t)runtime_len));
            if (binsize_is_int) {
                // the reference stores referenceFile['binsize'].item() unchanged: a Python int becomes an int64 array
                const int64_t bi = (int64_t)binsize;
                std::string m = npy_header("<i8", "()");
                m.append((const char *)&bi, 8);
                ok = ok && zw.add("binsize.npy", m);
            } else {
                ok = ok && zw.add("binsize.npy", npy_f64(&binsize, none));
            }
            ok = ok && zw.add("results_r.npy", npy_object_of_f64(r + (int64_t)i * row_stride, chrom_sizes, n_chrom), true);
            ok = ok && zw.add("results_z.npy", npy_object_of_f64(z + (int64_t)i * row_stride, chrom_sizes, n_chrom), true);
            ok = ok && zw.add("results_cwz.npy", npy_f64(cwz + (int64_t)i * n_sel, {(int64_t)n_sel}));
            const int nc = n_calls[i] < max_calls ? n_calls[i] : max_calls;
            // no calls: the reference stores np.array([]) -- float64, shape (0,)
            if (nc > 0) ok = ok && zw.add("results_calls.npy", npy_f64(calls + (int64_t)i * max_calls * 5, {(int64_t)nc, 5}));
            else ok = ok && zw.add("results_calls.npy", npy_f64(calls, {0}));
            ok = ok && zw.add("threshold_z.npy", npy_f64(&threshold_z, none));
            ok = ok && zw.add("asdef.npy", npy_f64(&asdef[i], none));
            const double aasdef = asdef[i] * threshold_z;
            ok = ok && zw.add("aasdef.npy", npy_f64(&aasdef, none));
            if (!ok || !zw.finish(out_paths[i])) rc = WC_NPZ_IO;
        } catch (...) {
            rc = WC_NPZ_IO;
        }
        status[i] = rc;
    });
    return WC_OK;
}

}  // extern "C"
