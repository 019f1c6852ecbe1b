// gguf_lite.cpp — see gguf_lite.h.  Container layout: SURVEY.md Appendix A.4.
#include "gguf_lite.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>

static const uint32_t GGUF_MAGIC = 0x46554747u;  // "GGUF" little-endian

gguf_file::~gguf_file() {
    if (map_addr) munmap(map_addr, map_size);
}
uint64_t gguf_file::get_u(const std::string & k, uint64_t def) const {
    auto it = kv.find(k);
    if (it == kv.end()) return def;
    if (it->second.type == GGUF_TYPE_FLOAT32 || it->second.type == GGUF_TYPE_FLOAT64) return (uint64_t) it->second.f;
    return it->second.u;
}
double gguf_file::get_f(const std::string & k, double def) const {
    auto it = kv.find(k);
    if (it == kv.end()) return def;
    if (it->second.type == GGUF_TYPE_FLOAT32 || it->second.type == GGUF_TYPE_FLOAT64) return it->second.f;
    return (double) it->second.u;
}
std::string gguf_file::get_s(const std::string & k, const std::string & def) const {
    auto it = kv.find(k);
    return it == kv.end() ? def : it->second.s;
}
const gguf_tensor_info * gguf_file::find(const std::string & name) const {
    for (auto & t : tensors)
        if (t.name == name) return &t;
    return nullptr;
}

namespace {
struct cursor {
    const uint8_t * p;
    const uint8_t * end;
    bool ok = true;
    template <typename T> T rd() {
        T v{};
        if (p + sizeof(T) > end) { ok = false; return v; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string rd_str() {
        uint64_t n = rd<uint64_t>();
        if (!ok || n > (uint64_t) (end - p)) { ok = false; return ""; }
        std::string s((const char *) p, (size_t) n);
        p += n;
        return s;
    }
};
size_t scalar_size(gguf_type t) {
    switch (t) {
        case GGUF_TYPE_UINT8: case GGUF_TYPE_INT8: case GGUF_TYPE_BOOL: return 1;
        case GGUF_TYPE_UINT16: case GGUF_TYPE_INT16: return 2;
        case GGUF_TYPE_UINT32: case GGUF_TYPE_INT32: case GGUF_TYPE_FLOAT32: return 4;
        case GGUF_TYPE_UINT64: case GGUF_TYPE_INT64: case GGUF_TYPE_FLOAT64: return 8;
        default: return 0;
    }
}
bool rd_scalar(cursor & c, gguf_type t, gguf_value & v) {
    switch (t) {
        case GGUF_TYPE_UINT8: v.u = c.rd<uint8_t>(); break;
        case GGUF_TYPE_INT8: v.u = (uint64_t) (int64_t) c.rd<int8_t>(); break;
        case GGUF_TYPE_UINT16: v.u = c.rd<uint16_t>(); break;
        case GGUF_TYPE_INT16: v.u = (uint64_t) (int64_t) c.rd<int16_t>(); break;
        case GGUF_TYPE_UINT32: v.u = c.rd<uint32_t>(); break;
        case GGUF_TYPE_INT32: v.u = (uint64_t) (int64_t) c.rd<int32_t>(); break;
        case GGUF_TYPE_FLOAT32: v.f = c.rd<float>(); break;
        case GGUF_TYPE_BOOL: v.u = c.rd<uint8_t>() != 0; break;
        case GGUF_TYPE_UINT64: v.u = c.rd<uint64_t>(); break;
        case GGUF_TYPE_INT64: v.u = (uint64_t) c.rd<int64_t>(); break;
        case GGUF_TYPE_FLOAT64: v.f = c.rd<double>(); break;
        case GGUF_TYPE_STRING: v.s = c.rd_str(); break;
        default: return false;
    }
    return c.ok;
}
}  // namespace

gguf_file * gguf_open(const char * path) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        fprintf(stderr, "gguf_open: cannot open %s\n", path);
        return nullptr;
    }
    struct stat st;
    fstat(fd, &st);
    void * addr = mmap(nullptr, (size_t) st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (addr == MAP_FAILED) {
        fprintf(stderr, "gguf_open: mmap failed for %s\n", path);
        return nullptr;
    }
    gguf_file * f = new gguf_file();
    f->map_addr = addr;
    f->map_size = (size_t) st.st_size;
    cursor c{(const uint8_t *) addr, (const uint8_t *) addr + st.st_size};
    const uint32_t magic = c.rd<uint32_t>();
    f->version = c.rd<uint32_t>();
    const uint64_t n_tensors = c.rd<uint64_t>();
    const uint64_t n_kv = c.rd<uint64_t>();
    if (!c.ok || magic != GGUF_MAGIC || (f->version != 3 && f->version != 2)) {
        fprintf(stderr, "gguf_open: %s: bad magic/version (magic=%08x version=%u)\n", path, magic, f->version);
        delete f;
        return nullptr;
    }
    for (uint64_t i = 0; i < n_kv && c.ok; ++i) {
        std::string key = c.rd_str();
        gguf_value v;
        v.type = (gguf_type) c.rd<uint32_t>();
        if (v.type == GGUF_TYPE_ARRAY) {
            v.arr_type = (gguf_type) c.rd<uint32_t>();
            v.arr_n = c.rd<uint64_t>();
            if (v.arr_type == GGUF_TYPE_STRING) {
                for (uint64_t k = 0; k < v.arr_n && c.ok; ++k) c.rd_str();
            } else {
                const size_t es = scalar_size(v.arr_type);
                if (es == 0 || v.arr_n > (uint64_t) (c.end - c.p) / es) c.ok = false;
                else c.p += es * v.arr_n;
            }
        } else if (!rd_scalar(c, v.type, v)) {
            c.ok = false;
        }
        f->kv[key] = v;
        f->kv_order.push_back(key);
    }
    f->alignment = (uint32_t) f->get_u("general.alignment", 32);
    for (uint64_t i = 0; i < n_tensors && c.ok; ++i) {
        gguf_tensor_info ti;
        ti.name = c.rd_str();
        ti.n_dims = (int) c.rd<uint32_t>();
        if (ti.n_dims < 1 || ti.n_dims > 4) { c.ok = false; break; }
        for (int d = 0; d < ti.n_dims; ++d) ti.ne[d] = (int64_t) c.rd<uint64_t>();
        ti.type = (ggml_type) c.rd<uint32_t>();
        ti.offset = c.rd<uint64_t>();
        if (ggml_abi_type_size(ti.type) == 0 || ti.ne[0] % ggml_abi_blck_size(ti.type) != 0) { c.ok = false; break; }
        ti.size = ggml_abi_row_size(ti.type, ti.ne[0]) * (size_t) (ti.ne[1] * ti.ne[2] * ti.ne[3]);
        f->tensors.push_back(ti);
    }
    if (!c.ok) {
        fprintf(stderr, "gguf_open: %s: truncated or malformed header\n", path);
        delete f;
        return nullptr;
    }
    const uint64_t pos = (uint64_t) (c.p - (const uint8_t *) addr);
    f->data_offset = (pos + f->alignment - 1) / f->alignment * f->alignment;
    for (auto & ti : f->tensors) {
        if (f->data_offset + ti.offset + ti.size > (uint64_t) st.st_size) {
            fprintf(stderr, "gguf_open: %s: tensor %s exceeds file size\n", path, ti.name.c_str());
            delete f;
            return nullptr;
        }
    }
    return f;
}

// ------------------------------------------------------------------------------------------------ writer
void gguf_writer::set_u32(const std::string & k, uint32_t v) {
    gguf_value x;
    x.type = GGUF_TYPE_UINT32;
    x.u = v;
    kvs.push_back({k, x});
}
void gguf_writer::set_f32(const std::string & k, float v) {
    gguf_value x;
    x.type = GGUF_TYPE_FLOAT32;
    x.f = v;
    kvs.push_back({k, x});
}
void gguf_writer::set_str(const std::string & k, const std::string & v) {
    gguf_value x;
    x.type = GGUF_TYPE_STRING;
    x.s = v;
    kvs.push_back({k, x});
}
void gguf_writer::add_tensor(const std::string & name, ggml_type type, int n_dims, const int64_t * ne) {
    gguf_tensor_info ti;
    ti.name = name;
    ti.type = type;
    ti.n_dims = n_dims;
    for (int i = 0; i < n_dims; ++i) ti.ne[i] = ne[i];
    ti.size = ggml_abi_row_size(type, ti.ne[0]) * (size_t) (ti.ne[1] * ti.ne[2] * ti.ne[3]);
    tensors.push_back(ti);
}

namespace {
void wr(FILE * f, const void * p, size_t n) { fwrite(p, 1, n, f); }
template <typename T> void wr_t(FILE * f, T v) { wr(f, &v, sizeof(T)); }
void wr_str(FILE * f, const std::string & s) {
    wr_t<uint64_t>(f, s.size());
    wr(f, s.data(), s.size());
}
}  // namespace

bool gguf_writer::write(const char * path, void (*fill)(const gguf_tensor_info & ti, void * dst, void * user), void * user) {
    const uint32_t alignment = 32;
    FILE * f = fopen(path, "wb");
    if (!f) return false;
    wr_t<uint32_t>(f, GGUF_MAGIC);
    wr_t<uint32_t>(f, 3);
    wr_t<uint64_t>(f, tensors.size());
    wr_t<uint64_t>(f, kvs.size());
    for (auto & kv : kvs) {
        wr_str(f, kv.first);
        wr_t<uint32_t>(f, kv.second.type);
        switch (kv.second.type) {
            case GGUF_TYPE_UINT32: wr_t<uint32_t>(f, (uint32_t) kv.second.u); break;
            case GGUF_TYPE_FLOAT32: wr_t<float>(f, (float) kv.second.f); break;
            case GGUF_TYPE_STRING: wr_str(f, kv.second.s); break;
            default: fclose(f); return false;
        }
    }
    uint64_t off = 0;
    for (auto & ti : tensors) {
        ti.offset = off;
        wr_str(f, ti.name);
        wr_t<uint32_t>(f, (uint32_t) ti.n_dims);
        for (int d = 0; d < ti.n_dims; ++d) wr_t<uint64_t>(f, (uint64_t) ti.ne[d]);
        wr_t<uint32_t>(f, (uint32_t) ti.type);
        wr_t<uint64_t>(f, ti.offset);
        off = (off + ti.size + alignment - 1) / alignment * alignment;
    }
    long pos = ftell(f);
    static const char zeros[64] = {0};
    size_t pad = (size_t) ((alignment - (pos % alignment)) % alignment);
    wr(f, zeros, pad);
    std::vector<char> buf;
    for (auto & ti : tensors) {
        buf.resize(ti.size);
        fill(ti, buf.data(), user);
        wr(f, buf.data(), ti.size);
        pad = (size_t) ((alignment - (ti.size % alignment)) % alignment);
        wr(f, zeros, pad);
    }
    bool ok = ferror(f) == 0;
    fclose(f);
    return ok;
}
