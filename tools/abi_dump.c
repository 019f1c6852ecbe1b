/*
 * tools/abi_dump.c — prints every struct size / field offset and every enum value that libggml-mi355x.so relies on.
 *
 * Why: include/ggml_abi.h restates ggml's backend ABI from memory (the llama.cpp submodule of the reference is empty,
 * SURVEY.md §0.1).  A maintainer with the real tree settles it in one diff:
 *
 *   gcc -DGGML_MAX_NAME=128 -I include                tools/abi_dump.c -o /tmp/abi_ours   && /tmp/abi_ours   > ours.txt
 *   gcc -DGGML_MAX_NAME=128 -DABI_DUMP_UPSTREAM \
 *       -I llama.cpp/ggml/include -I llama.cpp/ggml/src tools/abi_dump.c -o /tmp/abi_theirs && /tmp/abi_theirs > theirs.txt
 *   diff ours.txt theirs.txt        # empty = the library can be loaded by that host as built
 *
 * With -DABI_DUMP_UPSTREAM the same statements compile against upstream's ggml.h / ggml-backend.h / ggml-backend-impl.h /
 * ggml-impl.h / ggml-common.h instead of ggml_abi.h.  GENERATED from include/ggml_abi.h by the snippet in the commit that
 * added it (tests/test_host_and_abi.py::test_abi_dump_builds_and_is_complete checks it is in sync).
 */
#include <stddef.h>
#include <stdio.h>
#ifdef ABI_DUMP_UPSTREAM
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-backend-impl.h"
#include "ggml-impl.h"
#define GGML_COMMON_DECL_C
#include "ggml-common.h"
#else
#include "ggml_abi.h"
#endif

#define SZ(T) printf("sizeof(%s) = %zu\n", #T, sizeof(T))
#define OFF(T, f) printf("offsetof(%s, %s) = %zu\n", #T, #f, offsetof(T, f))
#define EN(e) printf("%s = %d\n", #e, (int) (e))

int main(void) {
    printf("GGML_MAX_DIMS = %d\nGGML_MAX_SRC = %d\nGGML_MAX_OP_PARAMS = %d\nGGML_MAX_NAME = %d\n", GGML_MAX_DIMS, GGML_MAX_SRC, GGML_MAX_OP_PARAMS, GGML_MAX_NAME);
    printf("GGML_BACKEND_API_VERSION = %d\n", GGML_BACKEND_API_VERSION);
    SZ(struct ggml_tensor);
    OFF(struct ggml_tensor, type);
    OFF(struct ggml_tensor, buffer);
    OFF(struct ggml_tensor, ne);
    OFF(struct ggml_tensor, nb);
    OFF(struct ggml_tensor, op);
    OFF(struct ggml_tensor, op_params);
    OFF(struct ggml_tensor, flags);
    OFF(struct ggml_tensor, src);
    OFF(struct ggml_tensor, view_src);
    OFF(struct ggml_tensor, view_offs);
    OFF(struct ggml_tensor, data);
    OFF(struct ggml_tensor, name);
    OFF(struct ggml_tensor, extra);
    OFF(struct ggml_tensor, padding);
    SZ(struct ggml_cgraph);
    OFF(struct ggml_cgraph, size);
    OFF(struct ggml_cgraph, n_nodes);
    OFF(struct ggml_cgraph, n_leafs);
    OFF(struct ggml_cgraph, nodes);
    OFF(struct ggml_cgraph, grads);
    OFF(struct ggml_cgraph, grad_accs);
    OFF(struct ggml_cgraph, leafs);
    OFF(struct ggml_cgraph, use_counts);
    OFF(struct ggml_cgraph, visited_hash_set);
    OFF(struct ggml_cgraph, order);
    SZ(struct ggml_hash_set);
    OFF(struct ggml_hash_set, size);
    OFF(struct ggml_hash_set, used);
    OFF(struct ggml_hash_set, keys);
    SZ(struct ggml_backend_reg);
    OFF(struct ggml_backend_reg, api_version);
    OFF(struct ggml_backend_reg, iface);
    OFF(struct ggml_backend_reg, context);
    SZ(struct ggml_backend_reg_i);
    OFF(struct ggml_backend_reg_i, get_name);
    OFF(struct ggml_backend_reg_i, get_device_count);
    OFF(struct ggml_backend_reg_i, get_device);
    OFF(struct ggml_backend_reg_i, get_proc_address);
    SZ(struct ggml_backend_device);
    OFF(struct ggml_backend_device, iface);
    OFF(struct ggml_backend_device, reg);
    OFF(struct ggml_backend_device, context);
    SZ(struct ggml_backend_device_i);
    OFF(struct ggml_backend_device_i, get_name);
    OFF(struct ggml_backend_device_i, get_description);
    OFF(struct ggml_backend_device_i, get_memory);
    OFF(struct ggml_backend_device_i, get_type);
    OFF(struct ggml_backend_device_i, get_props);
    OFF(struct ggml_backend_device_i, init_backend);
    OFF(struct ggml_backend_device_i, get_buffer_type);
    OFF(struct ggml_backend_device_i, get_host_buffer_type);
    OFF(struct ggml_backend_device_i, buffer_from_host_ptr);
    OFF(struct ggml_backend_device_i, supports_op);
    OFF(struct ggml_backend_device_i, supports_buft);
    OFF(struct ggml_backend_device_i, offload_op);
    OFF(struct ggml_backend_device_i, event_new);
    OFF(struct ggml_backend_device_i, event_free);
    OFF(struct ggml_backend_device_i, event_synchronize);
    SZ(struct ggml_backend_buffer_type);
    OFF(struct ggml_backend_buffer_type, iface);
    OFF(struct ggml_backend_buffer_type, device);
    OFF(struct ggml_backend_buffer_type, context);
    SZ(struct ggml_backend_buffer_type_i);
    OFF(struct ggml_backend_buffer_type_i, get_name);
    OFF(struct ggml_backend_buffer_type_i, alloc_buffer);
    OFF(struct ggml_backend_buffer_type_i, get_alignment);
    OFF(struct ggml_backend_buffer_type_i, get_max_size);
    OFF(struct ggml_backend_buffer_type_i, get_alloc_size);
    OFF(struct ggml_backend_buffer_type_i, is_host);
    SZ(struct ggml_backend_buffer);
    OFF(struct ggml_backend_buffer, iface);
    OFF(struct ggml_backend_buffer, buft);
    OFF(struct ggml_backend_buffer, context);
    OFF(struct ggml_backend_buffer, size);
    OFF(struct ggml_backend_buffer, usage);
    SZ(struct ggml_backend_buffer_i);
    OFF(struct ggml_backend_buffer_i, free_buffer);
    OFF(struct ggml_backend_buffer_i, get_base);
    OFF(struct ggml_backend_buffer_i, init_tensor);
    OFF(struct ggml_backend_buffer_i, memset_tensor);
    OFF(struct ggml_backend_buffer_i, set_tensor);
    OFF(struct ggml_backend_buffer_i, get_tensor);
    OFF(struct ggml_backend_buffer_i, cpy_tensor);
    OFF(struct ggml_backend_buffer_i, clear);
    OFF(struct ggml_backend_buffer_i, reset);
    SZ(struct ggml_backend);
    OFF(struct ggml_backend, guid);
    OFF(struct ggml_backend, iface);
    OFF(struct ggml_backend, device);
    OFF(struct ggml_backend, context);
    SZ(struct ggml_backend_i);
    OFF(struct ggml_backend_i, get_name);
    OFF(struct ggml_backend_i, free);
    OFF(struct ggml_backend_i, set_tensor_async);
    OFF(struct ggml_backend_i, get_tensor_async);
    OFF(struct ggml_backend_i, cpy_tensor_async);
    OFF(struct ggml_backend_i, synchronize);
    OFF(struct ggml_backend_i, graph_plan_create);
    OFF(struct ggml_backend_i, graph_plan_free);
    OFF(struct ggml_backend_i, graph_plan_update);
    OFF(struct ggml_backend_i, graph_plan_compute);
    OFF(struct ggml_backend_i, graph_compute);
    OFF(struct ggml_backend_i, event_record);
    OFF(struct ggml_backend_i, event_wait);
#if defined(ABI_DUMP_UPSTREAM) || GGML_ABI_HAS_GRAPH_OPTIMIZE
    OFF(struct ggml_backend_i, graph_optimize);
#endif
    SZ(struct ggml_backend_dev_props);
    OFF(struct ggml_backend_dev_props, name);
    OFF(struct ggml_backend_dev_props, description);
    OFF(struct ggml_backend_dev_props, memory_free);
    OFF(struct ggml_backend_dev_props, memory_total);
    OFF(struct ggml_backend_dev_props, type);
    OFF(struct ggml_backend_dev_props, caps);
    SZ(struct ggml_backend_dev_caps);
    OFF(struct ggml_backend_dev_caps, async);
    OFF(struct ggml_backend_dev_caps, host_buffer);
    OFF(struct ggml_backend_dev_caps, buffer_from_host_ptr);
    OFF(struct ggml_backend_dev_caps, events);
    SZ(struct ggml_backend_event);
    OFF(struct ggml_backend_event, device);
    OFF(struct ggml_backend_event, context);
    SZ(block_q8_0); OFF(block_q8_0, d); OFF(block_q8_0, qs);
    SZ(block_q4_K); OFF(block_q4_K, scales); OFF(block_q4_K, qs);
    SZ(block_q5_K); OFF(block_q5_K, scales); OFF(block_q5_K, qh); OFF(block_q5_K, qs);
    SZ(block_q6_K); OFF(block_q6_K, ql); OFF(block_q6_K, qh); OFF(block_q6_K, scales); OFF(block_q6_K, d);
    SZ(block_q8_K); OFF(block_q8_K, d); OFF(block_q8_K, qs); OFF(block_q8_K, bsums);
    /* enum ggml_type */
    EN(GGML_TYPE_F32);
    EN(GGML_TYPE_F16);
    EN(GGML_TYPE_Q4_0);
    EN(GGML_TYPE_Q4_1);
    EN(GGML_TYPE_Q5_0);
    EN(GGML_TYPE_Q5_1);
    EN(GGML_TYPE_Q8_0);
    EN(GGML_TYPE_Q8_1);
    EN(GGML_TYPE_Q2_K);
    EN(GGML_TYPE_Q3_K);
    EN(GGML_TYPE_Q4_K);
    EN(GGML_TYPE_Q5_K);
    EN(GGML_TYPE_Q6_K);
    EN(GGML_TYPE_Q8_K);
    EN(GGML_TYPE_IQ2_XXS);
    EN(GGML_TYPE_IQ2_XS);
    EN(GGML_TYPE_IQ3_XXS);
    EN(GGML_TYPE_IQ1_S);
    EN(GGML_TYPE_IQ4_NL);
    EN(GGML_TYPE_IQ3_S);
    EN(GGML_TYPE_IQ2_S);
    EN(GGML_TYPE_IQ4_XS);
    EN(GGML_TYPE_I8);
    EN(GGML_TYPE_I16);
    EN(GGML_TYPE_I32);
    EN(GGML_TYPE_I64);
    EN(GGML_TYPE_F64);
    EN(GGML_TYPE_IQ1_M);
    EN(GGML_TYPE_BF16);
    EN(GGML_TYPE_TQ1_0);
    EN(GGML_TYPE_TQ2_0);
    EN(GGML_TYPE_MXFP4);
    EN(GGML_TYPE_COUNT);
    /* enum ggml_op */
    EN(GGML_OP_NONE);
    EN(GGML_OP_DUP);
    EN(GGML_OP_ADD);
    EN(GGML_OP_ADD_ID);
    EN(GGML_OP_ADD1);
    EN(GGML_OP_ACC);
    EN(GGML_OP_SUB);
    EN(GGML_OP_MUL);
    EN(GGML_OP_DIV);
    EN(GGML_OP_SQR);
    EN(GGML_OP_SQRT);
    EN(GGML_OP_LOG);
    EN(GGML_OP_SIN);
    EN(GGML_OP_COS);
    EN(GGML_OP_SUM);
    EN(GGML_OP_SUM_ROWS);
    EN(GGML_OP_MEAN);
    EN(GGML_OP_ARGMAX);
    EN(GGML_OP_COUNT_EQUAL);
    EN(GGML_OP_REPEAT);
    EN(GGML_OP_REPEAT_BACK);
    EN(GGML_OP_CONCAT);
    EN(GGML_OP_SILU_BACK);
    EN(GGML_OP_NORM);
    EN(GGML_OP_RMS_NORM);
    EN(GGML_OP_RMS_NORM_BACK);
    EN(GGML_OP_GROUP_NORM);
    EN(GGML_OP_L2_NORM);
    EN(GGML_OP_MUL_MAT);
    EN(GGML_OP_MUL_MAT_ID);
    EN(GGML_OP_OUT_PROD);
    EN(GGML_OP_SCALE);
    EN(GGML_OP_SET);
    EN(GGML_OP_CPY);
    EN(GGML_OP_CONT);
    EN(GGML_OP_RESHAPE);
    EN(GGML_OP_VIEW);
    EN(GGML_OP_PERMUTE);
    EN(GGML_OP_TRANSPOSE);
    EN(GGML_OP_GET_ROWS);
    EN(GGML_OP_GET_ROWS_BACK);
    EN(GGML_OP_SET_ROWS);
    EN(GGML_OP_DIAG);
    EN(GGML_OP_DIAG_MASK_INF);
    EN(GGML_OP_DIAG_MASK_ZERO);
    EN(GGML_OP_SOFT_MAX);
    EN(GGML_OP_SOFT_MAX_BACK);
    EN(GGML_OP_ROPE);
    EN(GGML_OP_ROPE_BACK);
    EN(GGML_OP_CLAMP);
    EN(GGML_OP_CONV_TRANSPOSE_1D);
    EN(GGML_OP_IM2COL);
    EN(GGML_OP_IM2COL_BACK);
    EN(GGML_OP_CONV_2D);
    EN(GGML_OP_CONV_2D_DW);
    EN(GGML_OP_CONV_TRANSPOSE_2D);
    EN(GGML_OP_POOL_1D);
    EN(GGML_OP_POOL_2D);
    EN(GGML_OP_POOL_2D_BACK);
    EN(GGML_OP_UPSCALE);
    EN(GGML_OP_PAD);
    EN(GGML_OP_PAD_REFLECT_1D);
    EN(GGML_OP_ROLL);
    EN(GGML_OP_ARANGE);
    EN(GGML_OP_TIMESTEP_EMBEDDING);
    EN(GGML_OP_ARGSORT);
    EN(GGML_OP_LEAKY_RELU);
    EN(GGML_OP_FLASH_ATTN_EXT);
    EN(GGML_OP_FLASH_ATTN_BACK);
    EN(GGML_OP_SSM_CONV);
    EN(GGML_OP_SSM_SCAN);
    EN(GGML_OP_WIN_PART);
    EN(GGML_OP_WIN_UNPART);
    EN(GGML_OP_GET_REL_POS);
    EN(GGML_OP_ADD_REL_POS);
    EN(GGML_OP_RWKV_WKV6);
    EN(GGML_OP_GATED_LINEAR_ATTN);
    EN(GGML_OP_RWKV_WKV7);
    EN(GGML_OP_UNARY);
    EN(GGML_OP_MAP_CUSTOM1);
    EN(GGML_OP_MAP_CUSTOM2);
    EN(GGML_OP_MAP_CUSTOM3);
    EN(GGML_OP_CUSTOM);
    EN(GGML_OP_CROSS_ENTROPY_LOSS);
    EN(GGML_OP_CROSS_ENTROPY_LOSS_BACK);
    EN(GGML_OP_OPT_STEP_ADAMW);
    EN(GGML_OP_OPT_STEP_SGD);
    EN(GGML_OP_GLU);
    EN(GGML_OP_COUNT);
    /* enum ggml_unary_op */
    EN(GGML_UNARY_OP_ABS);
    EN(GGML_UNARY_OP_SGN);
    EN(GGML_UNARY_OP_NEG);
    EN(GGML_UNARY_OP_STEP);
    EN(GGML_UNARY_OP_TANH);
    EN(GGML_UNARY_OP_ELU);
    EN(GGML_UNARY_OP_RELU);
    EN(GGML_UNARY_OP_SIGMOID);
    EN(GGML_UNARY_OP_GELU);
    EN(GGML_UNARY_OP_GELU_QUICK);
    EN(GGML_UNARY_OP_SILU);
    EN(GGML_UNARY_OP_HARDSWISH);
    EN(GGML_UNARY_OP_HARDSIGMOID);
    EN(GGML_UNARY_OP_EXP);
    EN(GGML_UNARY_OP_GELU_ERF);
    EN(GGML_UNARY_OP_COUNT);
    /* enum ggml_glu_op */
    EN(GGML_GLU_OP_REGLU);
    EN(GGML_GLU_OP_GEGLU);
    EN(GGML_GLU_OP_SWIGLU);
    EN(GGML_GLU_OP_SWIGLU_OAI);
    EN(GGML_GLU_OP_GEGLU_ERF);
    EN(GGML_GLU_OP_GEGLU_QUICK);
    EN(GGML_GLU_OP_COUNT);
    /* enum ggml_tensor_flag */
    EN(GGML_TENSOR_FLAG_INPUT);
    EN(GGML_TENSOR_FLAG_OUTPUT);
    EN(GGML_TENSOR_FLAG_PARAM);
    EN(GGML_TENSOR_FLAG_LOSS);
    /* enum ggml_status */
    EN(GGML_STATUS_ALLOC_FAILED);
    EN(GGML_STATUS_FAILED);
    EN(GGML_STATUS_SUCCESS);
    EN(GGML_STATUS_ABORTED);
    /* enum ggml_backend_buffer_usage */
    EN(GGML_BACKEND_BUFFER_USAGE_ANY);
    EN(GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
    EN(GGML_BACKEND_BUFFER_USAGE_COMPUTE);
    /* enum ggml_backend_dev_type */
    EN(GGML_BACKEND_DEVICE_TYPE_CPU);
    EN(GGML_BACKEND_DEVICE_TYPE_GPU);
    EN(GGML_BACKEND_DEVICE_TYPE_ACCEL);
    /* enum ggml_prec */
    EN(GGML_PREC_DEFAULT);
    EN(GGML_PREC_F32);
    return 0;
}
