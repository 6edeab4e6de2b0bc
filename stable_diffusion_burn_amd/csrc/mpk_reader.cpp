// mpk_reader.cpp -- see mpk_reader.hpp.  A bounds-checked MessagePack walker over the memory-mapped record: tensors are
// indexed in place (name, shape, pointer into the mapping), nothing is copied until the engine stages them.
#include "mpk_reader.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>

#include "error.hpp"

namespace sdmi {

namespace {
struct Cur {
    const unsigned char* p;
    const unsigned char* end;
    const unsigned char* base;
};

[[noreturn]] void bad(const Cur& c, const char* what) {
    throw Error(SDMI_ERR_IO, std::string("mpk record: ") + what + " at byte " + std::to_string((size_t)(c.p - c.base)));
}
void need(const Cur& c, size_t n) {
    if ((size_t)(c.end - c.p) < n) bad(c, "truncated");
}
uint64_t be(Cur& c, int n) {
    need(c, (size_t)n);
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v = (v << 8) | c.p[i];
    c.p += n;
    return v;
}

enum Kind { K_NIL, K_BOOL, K_INT, K_FLOAT, K_STR, K_BIN, K_ARRAY, K_MAP, K_EXT };
struct Head {
    Kind kind;
    uint64_t n = 0;      // length (str / bin / ext payload) or element count (array / map)
    int64_t i = 0;       // K_INT / K_BOOL
    double f = 0;        // K_FLOAT
};

// reads one object header; for str / bin / ext the cursor is left at the payload
Head head(Cur& c) {
    need(c, 1);
    const unsigned b = *c.p++;
    Head h;
    if (b <= 0x7f) { h.kind = K_INT; h.i = b; return h; }
    if (b >= 0xe0) { h.kind = K_INT; h.i = (int8_t)b; return h; }
    if (b >= 0x80 && b <= 0x8f) { h.kind = K_MAP; h.n = b & 0x0f; return h; }
    if (b >= 0x90 && b <= 0x9f) { h.kind = K_ARRAY; h.n = b & 0x0f; return h; }
    if (b >= 0xa0 && b <= 0xbf) { h.kind = K_STR; h.n = b & 0x1f; return h; }
    switch (b) {
        case 0xc0: h.kind = K_NIL; return h;
        case 0xc2: h.kind = K_BOOL; h.i = 0; return h;
        case 0xc3: h.kind = K_BOOL; h.i = 1; return h;
        case 0xc4: h.kind = K_BIN; h.n = be(c, 1); return h;
        case 0xc5: h.kind = K_BIN; h.n = be(c, 2); return h;
        case 0xc6: h.kind = K_BIN; h.n = be(c, 4); return h;
        case 0xc7: h.kind = K_EXT; h.n = be(c, 1) + 1; return h;   // + the type byte
        case 0xc8: h.kind = K_EXT; h.n = be(c, 2) + 1; return h;
        case 0xc9: h.kind = K_EXT; h.n = be(c, 4) + 1; return h;
        case 0xca: { uint32_t u = (uint32_t)be(c, 4); float f; std::memcpy(&f, &u, 4); h.kind = K_FLOAT; h.f = f; return h; }
        case 0xcb: { uint64_t u = be(c, 8); double d; std::memcpy(&d, &u, 8); h.kind = K_FLOAT; h.f = d; return h; }
        case 0xcc: h.kind = K_INT; h.i = (int64_t)be(c, 1); return h;
        case 0xcd: h.kind = K_INT; h.i = (int64_t)be(c, 2); return h;
        case 0xce: h.kind = K_INT; h.i = (int64_t)be(c, 4); return h;
        case 0xcf: h.kind = K_INT; h.i = (int64_t)be(c, 8); return h;
        case 0xd0: h.kind = K_INT; h.i = (int8_t)be(c, 1); return h;
        case 0xd1: h.kind = K_INT; h.i = (int16_t)be(c, 2); return h;
        case 0xd2: h.kind = K_INT; h.i = (int32_t)be(c, 4); return h;
        case 0xd3: h.kind = K_INT; h.i = (int64_t)be(c, 8); return h;
        case 0xd4: h.kind = K_EXT; h.n = 2; return h;
        case 0xd5: h.kind = K_EXT; h.n = 3; return h;
        case 0xd6: h.kind = K_EXT; h.n = 5; return h;
        case 0xd7: h.kind = K_EXT; h.n = 9; return h;
        case 0xd8: h.kind = K_EXT; h.n = 17; return h;
        case 0xd9: h.kind = K_STR; h.n = be(c, 1); return h;
        case 0xda: h.kind = K_STR; h.n = be(c, 2); return h;
        case 0xdb: h.kind = K_STR; h.n = be(c, 4); return h;
        case 0xdc: h.kind = K_ARRAY; h.n = be(c, 2); return h;
        case 0xdd: h.kind = K_ARRAY; h.n = be(c, 4); return h;
        case 0xde: h.kind = K_MAP; h.n = be(c, 2); return h;
        case 0xdf: h.kind = K_MAP; h.n = be(c, 4); return h;
    }
    bad(c, "reserved type byte");
}

void skip(Cur& c, int depth = 0) {
    if (depth > 64) bad(c, "nesting too deep");
    const Head h = head(c);
    switch (h.kind) {
        case K_STR: case K_BIN: case K_EXT: need(c, h.n); c.p += h.n; break;
        case K_ARRAY: for (uint64_t i = 0; i < h.n; ++i) skip(c, depth + 1); break;
        case K_MAP: for (uint64_t i = 0; i < 2 * h.n; ++i) skip(c, depth + 1); break;
        default: break;
    }
}

std::string read_str(Cur& c) {
    const Head h = head(c);
    if (h.kind != K_STR && h.kind != K_BIN) bad(c, "expected a string");
    need(c, h.n);
    std::string s((const char*)c.p, (size_t)h.n);
    c.p += h.n;
    return s;
}
}  // namespace

struct MpkParser {
    MpkFile& f;
    Cur root;

    struct Entry { std::string key; const unsigned char* val; };

    std::vector<Entry> entries(Cur& c, uint64_t n) {
        std::vector<Entry> es;
        if (n > (uint64_t)(c.end - c.p)) bad(c, "map longer than the file");   // every entry takes at least two bytes
        es.reserve((size_t)n);
        for (uint64_t i = 0; i < n; ++i) {
            Cur k = c;
            const Head kh = head(k);
            std::string key;
            if (kh.kind == K_STR || kh.kind == K_BIN) { need(k, kh.n); key.assign((const char*)k.p, (size_t)kh.n); k.p += kh.n; }
            else if (kh.kind == K_INT) key = std::to_string(kh.i);
            else bad(c, "map key is neither a string nor an integer");
            c.p = k.p;
            es.push_back({key, c.p});
            skip(c);
        }
        return es;
    }
    static const Entry* find(const std::vector<Entry>& es, const char* key) {
        for (auto& e : es) if (e.key == key) return &e;
        return nullptr;
    }
    Cur at(const unsigned char* p) const { return Cur{p, root.end, root.base}; }

    static std::string dump_name(const std::vector<std::string>& path) {
        if (path.empty()) return "";
        if (path[0] == "alpha_cumulative_products") return "alphas_cumprod";
        std::vector<std::string> seg = path;
        if (seg[0] == "diffusion") seg[0] = "unet";
        if (seg.back() == "gamma") seg.back() = "weight";
        else if (seg.back() == "beta") seg.back() = "bias";
        if (seg.back() != "weight" && seg.back() != "bias") seg.push_back("weight");   // a bare Param field (clip.position_embedding)
        std::string s;
        for (size_t i = 0; i < seg.size(); ++i) { if (i) s += '/'; s += seg[i]; }
        return s;
    }

    std::vector<int64_t> read_shape(const unsigned char* p) {
        Cur c = at(p);
        const Head h = head(c);
        if (h.kind != K_ARRAY || h.n > 8) bad(c, "tensor shape is not a short array");
        std::vector<int64_t> shape;
        for (uint64_t i = 0; i < h.n; ++i) {
            const Head d = head(c);
            if (d.kind != K_INT || d.i < 0) bad(c, "tensor dimension is not a non-negative integer");
            shape.push_back(d.i);
        }
        return shape;
    }

    void tensor(const std::vector<Entry>& es, const std::vector<std::string>& path) {
        MpkTensor t;
        t.name = dump_name(path);
        t.shape = read_shape(find(es, "shape")->val);
        // checked product: a record is at most the file, so no honest tensor has more elements than the file has bytes -- a shape such as
        // [2^62 + 1] would otherwise wrap count * 4 to a small number and pass the byte-length check below with a 4-byte payload
        const size_t file_bytes = (size_t)(root.end - root.base);
        t.count = 1;
        for (int64_t d : t.shape) {
            if (d != 0 && t.count > file_bytes / (size_t)d) throw Error(SDMI_ERR_IO, "mpk record: tensor '" + t.name + "' has a shape larger than the file");
            t.count *= (size_t)d;
        }
        if (const Entry* dt = find(es, "dtype")) {
            Cur c = at(dt->val);
            Cur probe = c;
            const Head h = head(probe);
            std::string name;
            if (h.kind == K_STR) name = read_str(c);
            else if (h.kind == K_MAP && h.n == 1) { c = probe; name = read_str(c); }     // externally tagged enum {"F32": nil}
            else bad(c, "unrecognised dtype encoding");
            if (name != "F32") throw Error(SDMI_ERR_UNSUPPORTED, "mpk record: tensor '" + t.name + "' has dtype " + name +
                                                                     "; only NamedMpkFileRecorder<FullPrecisionSettings> (f32) records are supported");
        }
        if (const Entry* b = find(es, "bytes")) {
            Cur c = at(b->val);
            const Head h = head(c);
            if (h.kind == K_BIN || h.kind == K_STR) {
                need(c, h.n);
                if (h.n != t.count * 4) bad(c, "tensor byte length does not match its shape");
                t.data = c.p;
                t.file_offset = (size_t)(c.p - root.base);
            } else if (h.kind == K_ARRAY) {    // Vec<u8> written WITHOUT serde_bytes: an array of small integers
                if (h.n != t.count * 4 || h.n > (uint64_t)(root.end - c.p)) bad(c, "tensor byte length does not match its shape");   // (each element is >= 1 byte of file)
                f.owned_.emplace_back((size_t)t.count);
                unsigned char* dst = reinterpret_cast<unsigned char*>(f.owned_.back().data());
                for (uint64_t i = 0; i < h.n; ++i) {
                    const Head e = head(c);
                    if (e.kind != K_INT || e.i < 0 || e.i > 255) bad(c, "byte array element out of range");
                    dst[i] = (unsigned char)e.i;
                }
                t.data = dst;
                t.file_offset = 0;
            } else bad(c, "tensor bytes are neither bin nor an array");
        } else {   // burn <= 0.13 DataSerialize: "value": [f32, ...]
            Cur c = at(find(es, "value")->val);
            const Head h = head(c);
            if (h.kind != K_ARRAY || h.n != t.count || h.n > (uint64_t)(root.end - c.p)) bad(c, "tensor value count does not match its shape");
            f.owned_.emplace_back((size_t)t.count);
            float* dst = f.owned_.back().data();
            for (uint64_t i = 0; i < h.n; ++i) {
                const Head e = head(c);
                if (e.kind == K_FLOAT) dst[i] = (float)e.f;
                else if (e.kind == K_INT) dst[i] = (float)e.i;
                else bad(c, "tensor value is not a number");
            }
            t.data = reinterpret_cast<const unsigned char*>(dst);
            t.file_offset = 0;
        }
        f.tensors_.push_back(std::move(t));
    }

    void walk(const unsigned char* p, std::vector<std::string>& path, int depth) {
        if (depth > 48) throw Error(SDMI_ERR_IO, "mpk record: module nesting too deep");
        Cur c = at(p);
        const Head h = head(c);
        if (h.kind == K_MAP) {
            const std::vector<Entry> es = entries(c, h.n);
            if (find(es, "shape") && (find(es, "bytes") || find(es, "value"))) { tensor(es, path); return; }
            if (find(es, "param") && find(es, "id")) { walk(find(es, "param")->val, path, depth + 1); return; }   // ParamSerde
            for (auto& e : es) {
                path.push_back(e.key);
                walk(e.val, path, depth + 1);
                path.pop_back();
            }
        } else if (h.kind == K_ARRAY) {
            for (uint64_t i = 0; i < h.n; ++i) {
                path.push_back(std::to_string(i));
                const unsigned char* v = c.p;
                skip(c);
                walk(v, path, depth + 1);
                path.pop_back();
            }
        }
        // scalars / nil: constants and Option::None carry no tensor
    }
};

MpkFile::MpkFile(const std::string& path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw Error(SDMI_ERR_IO, "cannot open " + path);
    struct stat st;
    if (::fstat(fd_, &st) != 0 || st.st_size <= 0) { ::close(fd_); fd_ = -1; throw Error(SDMI_ERR_IO, "cannot stat (or empty file) " + path); }
    size_ = (size_t)st.st_size;
    map_ = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (map_ == MAP_FAILED) { map_ = nullptr; ::close(fd_); fd_ = -1; throw Error(SDMI_ERR_IO, "cannot map " + path); }
    try {
        const unsigned char* base = static_cast<const unsigned char*>(map_);
        MpkParser ps{*this, Cur{base, base + size_, base}};
        Cur c = ps.root;
        const Head h = head(c);
        std::vector<std::string> pathv;
        const unsigned char* item = base;
        if (h.kind == K_MAP) {
            const auto es = ps.entries(c, h.n);
            if (const auto* m = MpkParser::find(es, "metadata")) {
                Cur mc = ps.at(m->val);
                const Head mh = head(mc);
                if (mh.kind == K_MAP)
                    for (auto& e : ps.entries(mc, mh.n)) {
                        Cur vc = ps.at(e.val);
                        Cur probe = vc;
                        if (head(probe).kind != K_STR) continue;
                        if (e.key == "format") format_ = read_str(vc);
                        else if (e.key == "float") float_ = read_str(vc);
                    }
            }
            if (const auto* it = MpkParser::find(es, "item")) item = it->val;
        }
        if (!float_.empty() && float_ != "f32")
            throw Error(SDMI_ERR_UNSUPPORTED, "mpk record: metadata.float = " + float_ + "; only FullPrecisionSettings (f32) records are supported");
        ps.walk(item, pathv, 0);
        if (tensors_.empty()) throw Error(SDMI_ERR_IO, "mpk record: no tensors found in " + path);
    } catch (...) {
        ::munmap(map_, size_); map_ = nullptr;
        ::close(fd_); fd_ = -1;
        throw;
    }
}

MpkFile::~MpkFile() {
    if (map_) ::munmap(map_, size_);
    if (fd_ >= 0) ::close(fd_);
}

}  // namespace sdmi
