/*
 * gguf_lite.h — minimal GGUF v3 container reader/writer (own implementation; layout per SURVEY.md
 * Appendix A.4).  The reference reads GGUF through llama.cpp's gguf.cpp (un-vendored; touched by
 * llama-box/patches/llama.cpp/vocab.patch), which is out of scope to re-implement inside the backend;
 * this is the harness-side tool that fabricates and loads the synthetic models the parity tests and
 * bench.py run on (no real checkpoints exist offline).
 */
#ifndef GGUF_LITE_H
#define GGUF_LITE_H
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/ggml_abi.h"

enum gguf_type : uint32_t {
    GGUF_TYPE_UINT8 = 0, GGUF_TYPE_INT8 = 1, GGUF_TYPE_UINT16 = 2, GGUF_TYPE_INT16 = 3, GGUF_TYPE_UINT32 = 4,
    GGUF_TYPE_INT32 = 5, GGUF_TYPE_FLOAT32 = 6, GGUF_TYPE_BOOL = 7, GGUF_TYPE_STRING = 8, GGUF_TYPE_ARRAY = 9,
    GGUF_TYPE_UINT64 = 10, GGUF_TYPE_INT64 = 11, GGUF_TYPE_FLOAT64 = 12,
};

struct gguf_value {
    gguf_type type = GGUF_TYPE_UINT32;
    uint64_t u = 0;    // integer / bool payload
    double f = 0;      // float payload
    std::string s;     // string payload
    gguf_type arr_type = GGUF_TYPE_UINT32;
    uint64_t arr_n = 0;  // arrays are parsed for size only (tokenizer tables are not needed by the harness)
};

struct gguf_tensor_info {
    std::string name;
    ggml_type type = GGML_TYPE_F32;
    int n_dims = 0;
    int64_t ne[4] = {1, 1, 1, 1};
    uint64_t offset = 0;  // relative to data section
    size_t size = 0;
};

struct gguf_file {
    uint32_t version = 3;
    uint32_t alignment = 32;
    std::map<std::string, gguf_value> kv;
    std::vector<std::string> kv_order;
    std::vector<gguf_tensor_info> tensors;
    uint64_t data_offset = 0;  // absolute file offset of the data section
    // mapping (reader)
    void * map_addr = nullptr;
    size_t map_size = 0;

    ~gguf_file();
    bool has(const std::string & k) const { return kv.count(k) != 0; }
    uint64_t get_u(const std::string & k, uint64_t def = 0) const;
    double get_f(const std::string & k, double def = 0) const;
    std::string get_s(const std::string & k, const std::string & def = "") const;
    const gguf_tensor_info * find(const std::string & name) const;
    const void * tensor_data(const gguf_tensor_info & ti) const { return (const char *) map_addr + data_offset + ti.offset; }
};

// reader: mmap()s the file; returns nullptr (message on stderr) for bad magic/version/truncation
gguf_file * gguf_open(const char * path);

// writer: header + KVs + tensor infos are written first, then each tensor's bytes are streamed in order
struct gguf_writer {
    std::vector<std::pair<std::string, gguf_value>> kvs;
    std::vector<gguf_tensor_info> tensors;
    void set_u32(const std::string & k, uint32_t v);
    void set_f32(const std::string & k, float v);
    void set_str(const std::string & k, const std::string & v);
    void add_tensor(const std::string & name, ggml_type type, int n_dims, const int64_t * ne);
    // fill(ti, dst) must write ti.size bytes
    bool write(const char * path, void (*fill)(const gguf_tensor_info & ti, void * dst, void * user), void * user);
};
#endif
