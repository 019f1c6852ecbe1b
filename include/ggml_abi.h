/*
 * ggml_abi.h — restatement of the ggml public + backend-impl ABI that a ggml backend
 * shared library is compiled against.
 *
 * WHY THIS FILE EXISTS
 *   gpustack/llama-box drives every device through the ggml backend registry
 *   (/root/reference/llama-box/engine_param.hpp:542-545 -> ggml_backend_load_all();
 *   /root/reference/llama-box/patches/llama.cpp/pure_cpu.patch:80-102 lists the DL load order and the
 *   GGML_BACKEND_PATH hook).  The headers that define that interface (ggml.h, ggml-backend.h,
 *   ggml-backend-impl.h, ggml-impl.h) live in the llama.cpp submodule, which is NOT vendored in the
 *   reference snapshot (empty directory, no network).  This header is therefore a from-memory
 *   restatement of the ABI at the reference's vintage (llama.cpp ~Aug 2025; the registry in
 *   pure_cpu.patch lists webgpu but not zdnn), cross-checked against every in-tree use:
 *     - field list/order of ggml_tensor    : rpc_tensor mirror, llama-box/rpcserver.hpp:78-94, r/w at :702-785
 *     - GGML_MAX_NAME = 128                : /root/reference/CMakeLists.txt:62 (llama-box build define)
 *     - buffer->size / ->buft / ->iface    : llama-box/rpcserver.hpp:1073, :1466, :1421-1422
 *     - backend->device->iface.supports_op : llama-box/rpcserver.hpp:1530-1531
 *     - buffer_i / buft_i / backend_i sigs : llama-box/patches/llama.cpp/ggml-rpc.patch:224-274,:331
 *     - reg_i.get_proc_address(reg,name)   : llama-box/patches/llama.cpp/dynamic_link.patch:5-9
 *     - status enum                        : ggml-rpc.patch:266-273, llama-box/httpserver.hpp:3541-3545
 *
 *   It is UNVERIFIABLE offline.  Everything in this repository includes ONLY this header for ggml
 *   types, so that the first action when a real llama.cpp checkout is reachable is a single diff.
 *   Knobs that are known to have changed around the vintage are compile-time switches:
 *     GGML_MAX_NAME                 (128 for llama-box, 64 upstream default)
 *     GGML_BACKEND_API_VERSION      (1 at the vintage; 2 after the IGPU/device_id change, Sep 2025)
 *     GGML_ABI_HAS_GRAPH_OPTIMIZE   (trailing ggml_backend_i::graph_optimize, added Sep 2025)
 */
#ifndef GGML_ABI_H
#define GGML_ABI_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef GGML_MAX_NAME
#define GGML_MAX_NAME 128 /* /root/reference/CMakeLists.txt:62 */
#endif
#ifndef GGML_BACKEND_API_VERSION
#define GGML_BACKEND_API_VERSION 1
#endif
#ifndef GGML_ABI_HAS_GRAPH_OPTIMIZE
#define GGML_ABI_HAS_GRAPH_OPTIMIZE 0
#endif

#define GGML_MAX_DIMS 4
#define GGML_MAX_SRC 10
#define GGML_MAX_OP_PARAMS 64
#define GGML_MEM_ALIGN 16
#define GGML_DEFAULT_GRAPH_SIZE 2048
#define GGML_TENSOR_SIZE sizeof(struct ggml_tensor)

#define GGML_ROPE_TYPE_NEOX 2
#define GGML_ROPE_TYPE_MROPE 8
#define GGML_ROPE_TYPE_VISION 24

typedef uint16_t ggml_fp16_t;

enum ggml_status {
    GGML_STATUS_ALLOC_FAILED = -2,
    GGML_STATUS_FAILED = -1,
    GGML_STATUS_SUCCESS = 0,
    GGML_STATUS_ABORTED = 1,
};

/* numbering: SURVEY.md Appendix A.1 (rpcserver.hpp:697 range-checks type < GGML_TYPE_COUNT) */
enum ggml_type {
    GGML_TYPE_F32 = 0,
    GGML_TYPE_F16 = 1,
    GGML_TYPE_Q4_0 = 2,
    GGML_TYPE_Q4_1 = 3,
    GGML_TYPE_Q5_0 = 6,
    GGML_TYPE_Q5_1 = 7,
    GGML_TYPE_Q8_0 = 8,
    GGML_TYPE_Q8_1 = 9,
    GGML_TYPE_Q2_K = 10,
    GGML_TYPE_Q3_K = 11,
    GGML_TYPE_Q4_K = 12,
    GGML_TYPE_Q5_K = 13,
    GGML_TYPE_Q6_K = 14,
    GGML_TYPE_Q8_K = 15,
    GGML_TYPE_IQ2_XXS = 16,
    GGML_TYPE_IQ2_XS = 17,
    GGML_TYPE_IQ3_XXS = 18,
    GGML_TYPE_IQ1_S = 19,
    GGML_TYPE_IQ4_NL = 20,
    GGML_TYPE_IQ3_S = 21,
    GGML_TYPE_IQ2_S = 22,
    GGML_TYPE_IQ4_XS = 23,
    GGML_TYPE_I8 = 24,
    GGML_TYPE_I16 = 25,
    GGML_TYPE_I32 = 26,
    GGML_TYPE_I64 = 27,
    GGML_TYPE_F64 = 28,
    GGML_TYPE_IQ1_M = 29,
    GGML_TYPE_BF16 = 30,
    GGML_TYPE_TQ1_0 = 34,
    GGML_TYPE_TQ2_0 = 35,
    GGML_TYPE_MXFP4 = 39,
    GGML_TYPE_COUNT = 40,
};

enum ggml_prec {
    GGML_PREC_DEFAULT = 0,
    GGML_PREC_F32 = 10,
};

/*
 * Operator numbering at the vintage (ADD_ID present since the gpt-oss merge, Aug 2025).  The numbers
 * are from memory; our own driver and tests only ever use these symbolic names, so a renumbering at
 * integration time is a recompile, not a code change (SURVEY.md Appendix B "unknowns").
 */
enum ggml_op {
    GGML_OP_NONE = 0,
    GGML_OP_DUP,
    GGML_OP_ADD,
    GGML_OP_ADD_ID,
    GGML_OP_ADD1,
    GGML_OP_ACC,
    GGML_OP_SUB,
    GGML_OP_MUL,
    GGML_OP_DIV,
    GGML_OP_SQR,
    GGML_OP_SQRT,
    GGML_OP_LOG,
    GGML_OP_SIN,
    GGML_OP_COS,
    GGML_OP_SUM,
    GGML_OP_SUM_ROWS,
    GGML_OP_MEAN,
    GGML_OP_ARGMAX,
    GGML_OP_COUNT_EQUAL,
    GGML_OP_REPEAT,
    GGML_OP_REPEAT_BACK,
    GGML_OP_CONCAT,
    GGML_OP_SILU_BACK,
    GGML_OP_NORM,
    GGML_OP_RMS_NORM,
    GGML_OP_RMS_NORM_BACK,
    GGML_OP_GROUP_NORM,
    GGML_OP_L2_NORM,
    GGML_OP_MUL_MAT,
    GGML_OP_MUL_MAT_ID,
    GGML_OP_OUT_PROD,
    GGML_OP_SCALE,
    GGML_OP_SET,
    GGML_OP_CPY,
    GGML_OP_CONT,
    GGML_OP_RESHAPE,
    GGML_OP_VIEW,
    GGML_OP_PERMUTE,
    GGML_OP_TRANSPOSE,
    GGML_OP_GET_ROWS,
    GGML_OP_GET_ROWS_BACK,
    GGML_OP_SET_ROWS,
    GGML_OP_DIAG,
    GGML_OP_DIAG_MASK_INF,
    GGML_OP_DIAG_MASK_ZERO,
    GGML_OP_SOFT_MAX,
    GGML_OP_SOFT_MAX_BACK,
    GGML_OP_ROPE,
    GGML_OP_ROPE_BACK,
    GGML_OP_CLAMP,
    GGML_OP_CONV_TRANSPOSE_1D,
    GGML_OP_IM2COL,
    GGML_OP_IM2COL_BACK,
    GGML_OP_CONV_2D,
    GGML_OP_CONV_2D_DW,
    GGML_OP_CONV_TRANSPOSE_2D,
    GGML_OP_POOL_1D,
    GGML_OP_POOL_2D,
    GGML_OP_POOL_2D_BACK,
    GGML_OP_UPSCALE,
    GGML_OP_PAD,
    GGML_OP_PAD_REFLECT_1D,
    GGML_OP_ROLL,
    GGML_OP_ARANGE,
    GGML_OP_TIMESTEP_EMBEDDING,
    GGML_OP_ARGSORT,
    GGML_OP_LEAKY_RELU,
    GGML_OP_FLASH_ATTN_EXT,
    GGML_OP_FLASH_ATTN_BACK,
    GGML_OP_SSM_CONV,
    GGML_OP_SSM_SCAN,
    GGML_OP_WIN_PART,
    GGML_OP_WIN_UNPART,
    GGML_OP_GET_REL_POS,
    GGML_OP_ADD_REL_POS,
    GGML_OP_RWKV_WKV6,
    GGML_OP_GATED_LINEAR_ATTN,
    GGML_OP_RWKV_WKV7,
    GGML_OP_UNARY,
    GGML_OP_MAP_CUSTOM1,
    GGML_OP_MAP_CUSTOM2,
    GGML_OP_MAP_CUSTOM3,
    GGML_OP_CUSTOM,
    GGML_OP_CROSS_ENTROPY_LOSS,
    GGML_OP_CROSS_ENTROPY_LOSS_BACK,
    GGML_OP_OPT_STEP_ADAMW,
    GGML_OP_OPT_STEP_SGD,
    GGML_OP_GLU,
    GGML_OP_COUNT,
};

enum ggml_unary_op {
    GGML_UNARY_OP_ABS = 0,
    GGML_UNARY_OP_SGN,
    GGML_UNARY_OP_NEG,
    GGML_UNARY_OP_STEP,
    GGML_UNARY_OP_TANH,
    GGML_UNARY_OP_ELU,
    GGML_UNARY_OP_RELU,
    GGML_UNARY_OP_SIGMOID,
    GGML_UNARY_OP_GELU,
    GGML_UNARY_OP_GELU_QUICK,
    GGML_UNARY_OP_SILU,
    GGML_UNARY_OP_HARDSWISH,
    GGML_UNARY_OP_HARDSIGMOID,
    GGML_UNARY_OP_EXP,
    GGML_UNARY_OP_GELU_ERF,
    GGML_UNARY_OP_COUNT,
};

enum ggml_glu_op {
    GGML_GLU_OP_REGLU = 0,
    GGML_GLU_OP_GEGLU,
    GGML_GLU_OP_SWIGLU,
    GGML_GLU_OP_SWIGLU_OAI,
    GGML_GLU_OP_GEGLU_ERF,
    GGML_GLU_OP_GEGLU_QUICK,
    GGML_GLU_OP_COUNT,
};

enum ggml_tensor_flag {
    GGML_TENSOR_FLAG_INPUT = 1,
    GGML_TENSOR_FLAG_OUTPUT = 2,
    GGML_TENSOR_FLAG_PARAM = 4,
    GGML_TENSOR_FLAG_LOSS = 8,
};

struct ggml_backend_buffer;

/* field order corroborated by rpc_tensor (llama-box/rpcserver.hpp:78-94) */
struct ggml_tensor {
    enum ggml_type type;
    struct ggml_backend_buffer * buffer;
    int64_t ne[GGML_MAX_DIMS];
    size_t nb[GGML_MAX_DIMS];
    enum ggml_op op;
    int32_t op_params[GGML_MAX_OP_PARAMS / sizeof(int32_t)];
    int32_t flags;
    struct ggml_tensor * src[GGML_MAX_SRC];
    struct ggml_tensor * view_src;
    size_t view_offs;
    void * data;
    char name[GGML_MAX_NAME];
    void * extra;
    char padding[8];
};

/* ---- graph (ggml-impl.h); only size/n_nodes/n_leafs/nodes/leafs are read by a backend ---- */
typedef uint32_t ggml_bitset_t;
struct ggml_hash_set {
    size_t size;
    ggml_bitset_t * used;
    struct ggml_tensor ** keys;
};
enum ggml_cgraph_eval_order {
    GGML_CGRAPH_EVAL_ORDER_LEFT_TO_RIGHT = 0,
    GGML_CGRAPH_EVAL_ORDER_RIGHT_TO_LEFT,
    GGML_CGRAPH_EVAL_ORDER_COUNT
};
struct ggml_cgraph {
    int size;
    int n_nodes; /* used at llama-box/rpcserver.hpp:1374-1384 */
    int n_leafs;
    struct ggml_tensor ** nodes;
    struct ggml_tensor ** grads;
    struct ggml_tensor ** grad_accs;
    struct ggml_tensor ** leafs;
    int32_t * use_counts;
    struct ggml_hash_set visited_hash_set;
    enum ggml_cgraph_eval_order order;
};

/* ---------------------------------- backend interface ---------------------------------- */
typedef struct ggml_backend_buffer_type * ggml_backend_buffer_type_t;
typedef struct ggml_backend_buffer * ggml_backend_buffer_t;
typedef struct ggml_backend_event * ggml_backend_event_t;
typedef struct ggml_backend * ggml_backend_t;
typedef void * ggml_backend_graph_plan_t;
typedef struct ggml_backend_reg * ggml_backend_reg_t;
typedef struct ggml_backend_device * ggml_backend_dev_t;
typedef uint8_t ggml_guid[16];
typedef ggml_guid * ggml_guid_t;

enum ggml_backend_buffer_usage {
    GGML_BACKEND_BUFFER_USAGE_ANY = 0,
    GGML_BACKEND_BUFFER_USAGE_WEIGHTS = 1,
    GGML_BACKEND_BUFFER_USAGE_COMPUTE = 2,
};

enum ggml_backend_dev_type {
    GGML_BACKEND_DEVICE_TYPE_CPU,
    GGML_BACKEND_DEVICE_TYPE_GPU,
    GGML_BACKEND_DEVICE_TYPE_ACCEL,
};

struct ggml_backend_dev_caps {
    bool async;
    bool host_buffer;
    bool buffer_from_host_ptr;
    bool events;
};

struct ggml_backend_dev_props {
    const char * name;
    const char * description;
    size_t memory_free;
    size_t memory_total;
    enum ggml_backend_dev_type type;
    struct ggml_backend_dev_caps caps;
};

struct ggml_backend_buffer_type_i {
    const char * (*get_name)(ggml_backend_buffer_type_t buft);
    ggml_backend_buffer_t (*alloc_buffer)(ggml_backend_buffer_type_t buft, size_t size);
    size_t (*get_alignment)(ggml_backend_buffer_type_t buft);
    size_t (*get_max_size)(ggml_backend_buffer_type_t buft);                                   /* optional */
    size_t (*get_alloc_size)(ggml_backend_buffer_type_t buft, const struct ggml_tensor * t);  /* optional */
    bool (*is_host)(ggml_backend_buffer_type_t buft);                                          /* optional */
};
struct ggml_backend_buffer_type {
    struct ggml_backend_buffer_type_i iface;
    ggml_backend_dev_t device;
    void * context;
};

struct ggml_backend_buffer_i {
    void (*free_buffer)(ggml_backend_buffer_t buffer);
    void * (*get_base)(ggml_backend_buffer_t buffer);
    enum ggml_status (*init_tensor)(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor); /* optional */
    void (*memset_tensor)(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor, uint8_t value, size_t offset, size_t size);
    void (*set_tensor)(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size);
    void (*get_tensor)(ggml_backend_buffer_t buffer, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size);
    bool (*cpy_tensor)(ggml_backend_buffer_t buffer, const struct ggml_tensor * src, struct ggml_tensor * dst); /* optional */
    void (*clear)(ggml_backend_buffer_t buffer, uint8_t value);
    void (*reset)(ggml_backend_buffer_t buffer);                                                /* optional */
};
struct ggml_backend_buffer {
    struct ggml_backend_buffer_i iface;
    ggml_backend_buffer_type_t buft;
    void * context;
    size_t size;
    enum ggml_backend_buffer_usage usage;
};

struct ggml_backend_i {
    const char * (*get_name)(ggml_backend_t backend);
    void (*free)(ggml_backend_t backend);
    void (*set_tensor_async)(ggml_backend_t backend, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size);
    void (*get_tensor_async)(ggml_backend_t backend, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size);
    bool (*cpy_tensor_async)(ggml_backend_t backend_src, ggml_backend_t backend_dst, const struct ggml_tensor * src, struct ggml_tensor * dst);
    void (*synchronize)(ggml_backend_t backend);
    ggml_backend_graph_plan_t (*graph_plan_create)(ggml_backend_t backend, const struct ggml_cgraph * cgraph);
    void (*graph_plan_free)(ggml_backend_t backend, ggml_backend_graph_plan_t plan);
    void (*graph_plan_update)(ggml_backend_t backend, ggml_backend_graph_plan_t plan, const struct ggml_cgraph * cgraph);
    enum ggml_status (*graph_plan_compute)(ggml_backend_t backend, ggml_backend_graph_plan_t plan);
    enum ggml_status (*graph_compute)(ggml_backend_t backend, struct ggml_cgraph * cgraph);
    void (*event_record)(ggml_backend_t backend, ggml_backend_event_t event);
    void (*event_wait)(ggml_backend_t backend, ggml_backend_event_t event);
#if GGML_ABI_HAS_GRAPH_OPTIMIZE
    void (*graph_optimize)(ggml_backend_t backend, struct ggml_cgraph * cgraph);
#endif
};
struct ggml_backend {
    ggml_guid_t guid;
    struct ggml_backend_i iface;
    ggml_backend_dev_t device;
    void * context;
};
struct ggml_backend_event {
    struct ggml_backend_device * device;
    void * context;
};

struct ggml_backend_device_i {
    const char * (*get_name)(ggml_backend_dev_t dev);
    const char * (*get_description)(ggml_backend_dev_t dev);
    void (*get_memory)(ggml_backend_dev_t dev, size_t * free, size_t * total);
    enum ggml_backend_dev_type (*get_type)(ggml_backend_dev_t dev);
    void (*get_props)(ggml_backend_dev_t dev, struct ggml_backend_dev_props * props);
    ggml_backend_t (*init_backend)(ggml_backend_dev_t dev, const char * params);
    ggml_backend_buffer_type_t (*get_buffer_type)(ggml_backend_dev_t dev);
    ggml_backend_buffer_type_t (*get_host_buffer_type)(ggml_backend_dev_t dev);                       /* optional */
    ggml_backend_buffer_t (*buffer_from_host_ptr)(ggml_backend_dev_t dev, void * ptr, size_t size, size_t max_tensor_size); /* optional */
    bool (*supports_op)(ggml_backend_dev_t dev, const struct ggml_tensor * op);
    bool (*supports_buft)(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft);
    bool (*offload_op)(ggml_backend_dev_t dev, const struct ggml_tensor * op);                        /* optional */
    ggml_backend_event_t (*event_new)(ggml_backend_dev_t dev);                                        /* optional */
    void (*event_free)(ggml_backend_dev_t dev, ggml_backend_event_t event);                           /* optional */
    void (*event_synchronize)(ggml_backend_dev_t dev, ggml_backend_event_t event);                    /* optional */
};
struct ggml_backend_device {
    struct ggml_backend_device_i iface;
    ggml_backend_reg_t reg;
    void * context;
};

struct ggml_backend_reg_i {
    const char * (*get_name)(ggml_backend_reg_t reg);
    size_t (*get_device_count)(ggml_backend_reg_t reg);
    ggml_backend_dev_t (*get_device)(ggml_backend_reg_t reg, size_t index);
    void * (*get_proc_address)(ggml_backend_reg_t reg, const char * name);                            /* optional */
};
struct ggml_backend_reg {
    int api_version;
    struct ggml_backend_reg_i iface;
    void * context;
};

/* dynamic-loading entry points a backend .so exports (what GGML_BACKEND_DL_IMPL expands to) */
typedef ggml_backend_reg_t (*ggml_backend_init_t)(void);
typedef int (*ggml_backend_score_t)(void);

/* feature list returned through get_proc_address("ggml_backend_get_features") */
struct ggml_backend_feature {
    const char * name;
    const char * value;
};
typedef struct ggml_backend_feature * (*ggml_backend_get_features_t)(ggml_backend_reg_t reg);
/* row-split hook: get_proc_address("ggml_backend_split_buffer_type") */
typedef ggml_backend_buffer_type_t (*ggml_backend_split_buffer_type_t)(int main_device, const float * tensor_split);

/* ------------------------- quantised block formats (ggml-common.h) ------------------------- */
#define QK_K 256
#define K_SCALE_SIZE 12
#define QK8_0 32

#pragma pack(push, 1)
typedef struct { ggml_fp16_t d; int8_t qs[QK8_0]; } block_q8_0;                                            /* 34 B */
/* the legacy 32-value formats a KV cache may be kept in (-ctk / -ctv: llama-box/engine_param.hpp:51-54) and the Q8_1 activation block their dots take */
typedef struct { ggml_fp16_t d; uint8_t qs[16]; } block_q4_0;                                              /* 18 B */
typedef struct { ggml_fp16_t d; ggml_fp16_t m; uint8_t qs[16]; } block_q4_1;                               /* 20 B */
typedef struct { ggml_fp16_t d; uint8_t qh[4]; uint8_t qs[16]; } block_q5_0;                               /* 22 B */
typedef struct { ggml_fp16_t d; ggml_fp16_t m; uint8_t qh[4]; uint8_t qs[16]; } block_q5_1;                /* 24 B */
typedef struct { ggml_fp16_t d; ggml_fp16_t s; int8_t qs[32]; } block_q8_1;                                /* 36 B: s = d * sum(qs) */
typedef struct { ggml_fp16_t d; uint8_t qs[16]; } block_iq4_nl;                                            /* 18 B */
typedef struct { ggml_fp16_t d; ggml_fp16_t dmin; uint8_t scales[K_SCALE_SIZE]; uint8_t qs[QK_K / 2]; } block_q4_K;   /* 144 B */
typedef struct { ggml_fp16_t d; ggml_fp16_t dmin; uint8_t scales[K_SCALE_SIZE]; uint8_t qh[QK_K / 8]; uint8_t qs[QK_K / 2]; } block_q5_K; /* 176 B */
typedef struct { uint8_t ql[QK_K / 2]; uint8_t qh[QK_K / 4]; int8_t scales[QK_K / 16]; ggml_fp16_t d; } block_q6_K;   /* 210 B */
typedef struct { float d; int8_t qs[QK_K]; int16_t bsums[QK_K / 16]; } block_q8_K;                        /* 292 B */
#pragma pack(pop)

#ifdef __cplusplus
static_assert(sizeof(block_q8_0) == 34, "q8_0");
static_assert(sizeof(block_q4_0) == 18 && sizeof(block_q4_1) == 20 && sizeof(block_q5_0) == 22 && sizeof(block_q5_1) == 24 && sizeof(block_q8_1) == 36 && sizeof(block_iq4_nl) == 18, "legacy 32-value blocks");
static_assert(sizeof(block_q4_K) == 144, "q4_K");
static_assert(sizeof(block_q5_K) == 176, "q5_K");
static_assert(sizeof(block_q6_K) == 210, "q6_K");
static_assert(sizeof(block_q8_K) == 292, "q8_K");
static_assert(sizeof(struct ggml_tensor) == 272 + GGML_MAX_NAME, "ggml_tensor layout");
static_assert(offsetof(struct ggml_tensor, op) == 80, "ggml_tensor.op");
static_assert(offsetof(struct ggml_tensor, src) == 152, "ggml_tensor.src");
static_assert(offsetof(struct ggml_tensor, data) == 248, "ggml_tensor.data");
static_assert(sizeof(struct ggml_tensor) % GGML_MEM_ALIGN == 0, "ggml_tensor align");
#endif

/* ------- small inline helpers every side needs (own implementations, not ggml's code) ------- */
static inline int64_t ggml_abi_blck_size(enum ggml_type t) {
    switch (t) {
        case GGML_TYPE_Q8_0: case GGML_TYPE_Q4_0: case GGML_TYPE_Q4_1: case GGML_TYPE_Q5_0: case GGML_TYPE_Q5_1: case GGML_TYPE_Q8_1: case GGML_TYPE_IQ4_NL: return 32;
        case GGML_TYPE_Q4_K: case GGML_TYPE_Q5_K: case GGML_TYPE_Q6_K: case GGML_TYPE_Q8_K: return 256;
        default: return 1;
    }
}
static inline size_t ggml_abi_type_size(enum ggml_type t) {
    switch (t) {
        case GGML_TYPE_F32: case GGML_TYPE_I32: return 4;
        case GGML_TYPE_F16: case GGML_TYPE_BF16: case GGML_TYPE_I16: return 2;
        case GGML_TYPE_I8: return 1;
        case GGML_TYPE_I64: case GGML_TYPE_F64: return 8;
        case GGML_TYPE_Q8_0: return 34;
        case GGML_TYPE_Q4_0: case GGML_TYPE_IQ4_NL: return 18;
        case GGML_TYPE_Q4_1: return 20;
        case GGML_TYPE_Q5_0: return 22;
        case GGML_TYPE_Q5_1: return 24;
        case GGML_TYPE_Q8_1: return 36;
        case GGML_TYPE_Q4_K: return 144;
        case GGML_TYPE_Q5_K: return 176;
        case GGML_TYPE_Q6_K: return 210;
        case GGML_TYPE_Q8_K: return 292;
        default: return 0; /* unsupported here */
    }
}
static inline size_t ggml_abi_row_size(enum ggml_type t, int64_t ne) {
    return (size_t) (ne / ggml_abi_blck_size(t)) * ggml_abi_type_size(t);
}
static inline int64_t ggml_abi_nelements(const struct ggml_tensor * t) {
    return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3];
}
static inline int64_t ggml_abi_nrows(const struct ggml_tensor * t) { return t->ne[1] * t->ne[2] * t->ne[3]; }
/* bytes spanned by the tensor (same definition as ggml_nbytes) */
static inline size_t ggml_abi_nbytes(const struct ggml_tensor * t) {
    for (int i = 0; i < GGML_MAX_DIMS; ++i) if (t->ne[i] <= 0) return 0;
    const int64_t blck = ggml_abi_blck_size(t->type);
    size_t n;
    if (blck == 1) {
        n = ggml_abi_type_size(t->type);
        for (int i = 0; i < GGML_MAX_DIMS; ++i) n += (size_t) (t->ne[i] - 1) * t->nb[i];
    } else {
        n = (size_t) t->ne[0] * t->nb[0] / (size_t) blck;
        for (int i = 1; i < GGML_MAX_DIMS; ++i) n += (size_t) (t->ne[i] - 1) * t->nb[i];
    }
    return n;
}
static inline bool ggml_abi_is_contiguous(const struct ggml_tensor * t) {
    size_t expect = ggml_abi_type_size(t->type);
    if (t->ne[0] != ggml_abi_blck_size(t->type) && t->nb[0] != expect) return false;
    expect = expect * (size_t) (t->ne[0] / ggml_abi_blck_size(t->type));
    for (int i = 1; i < GGML_MAX_DIMS; ++i) {
        if (t->ne[i] != 1 && t->nb[i] != expect) return false;
        expect *= (size_t) t->ne[i];
    }
    return true;
}
static inline float ggml_abi_op_param_f32(const struct ggml_tensor * t, int i) {
    float v; __builtin_memcpy(&v, &t->op_params[i], 4); return v;
}

#ifdef __cplusplus
}
#endif
#endif /* GGML_ABI_H */
