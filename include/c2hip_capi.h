/*
 * c2hip_capi.h -- C-ABI of the Caffe2-shaped operator surface for HIPContext.
 *
 * These are the entry points the reference's host side binds for this path.
 * In the reference the binding is pybind11 over C++ (caffe2/python/
 * pybind_state.cc): workspace.FeedBlob / FetchBlob / RunOperatorOnce /
 * CreateNet+RunNet, core.GetGradientForOp, dyndep.InitOpsLibrary +
 * RefreshRegisteredOperators.  Each function below names the call it
 * replaces; operator definitions cross the boundary as the SAME protobuf wire
 * bytes (`OperatorDef.SerializeToString()`, caffe2/proto/caffe2.proto:142-172)
 * the reference passes, decoded by a built-in codec (no protobuf dependency).
 *
 * Everything is plain C: opaque handles, pointers and sizes.  Functions
 * return 0 on success and non-zero on failure; c2hip_last_error() then holds
 * the EnforceNotMet text (thread-local), as the reference surfaces C++
 * exceptions to Python (caffe2/core/operator.h:369-395).
 */
#ifndef C2HIP_CAPI_H_
#define C2HIP_CAPI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C2HIP_CAPI __attribute__((visibility("default")))

/* device types (caffe2.proto:112-116; HIP = 6 as in later upstream Caffe2).
 * A definition that says CUDA is served by the HIP registry. */
#define C2HIP_CPU 0
#define C2HIP_CUDA 1
#define C2HIP_HIP 6

/* TensorProto::DataType ids (caffe2.proto:33-49) */
#define C2HIP_FLOAT 1
#define C2HIP_INT32 2
#define C2HIP_INT64 10
#define C2HIP_DOUBLE 13

#define C2HIP_MAX_DIMS 8

typedef struct c2hip_workspace c2hip_workspace;
typedef struct c2hip_operator c2hip_operator;

C2HIP_CAPI const char* c2hip_last_error(void);

/* --- workspace (caffe2/core/workspace.h:63-300; pybind_state.cc
 *     "Workspace", "create_blob", "has_blob", "blobs") --------------------- */
C2HIP_CAPI c2hip_workspace* c2hip_workspace_create(void);
C2HIP_CAPI void c2hip_workspace_destroy(c2hip_workspace* ws);
C2HIP_CAPI int c2hip_has_blob(c2hip_workspace* ws, const char* name);
C2HIP_CAPI int c2hip_remove_blob(c2hip_workspace* ws, const char* name);
/* writes the blob names separated by '\n'; returns the byte count needed */
C2HIP_CAPI size_t c2hip_blobs(c2hip_workspace* ws, char* buf, size_t buflen);

/* --- FeedBlob / FetchBlob (pybind_state.cc feed_blob / fetch_blob;
 *     detectron/lib/roi_data/loader.py:250-274 feeds this way) -------------
 * Copies a host array into a tensor blob on (device_type, device_id). */
C2HIP_CAPI int c2hip_feed_blob(c2hip_workspace* ws, const char* name, const void* host_data,
                               const int64_t* dims, int ndim, int dtype, int device_type,
                               int device_id);
/* Shape / type of a tensor blob; dims must hold C2HIP_MAX_DIMS entries. */
C2HIP_CAPI int c2hip_blob_info(c2hip_workspace* ws, const char* name, int* dtype,
                               int* device_type, int* ndim, int64_t* dims);
/* Copies the tensor to host memory (nbytes must match). */
C2HIP_CAPI int c2hip_fetch_blob(c2hip_workspace* ws, const char* name, void* host_out,
                                size_t nbytes);
/* Device pointer of a HIP tensor blob (NULL + error if absent / CPU).  The pointer is writable: the call counts
 * as a write to the blob (derived data cached by operators, e.g. packed filters, is rebuilt on their next run);
 * writes made LATER through a pointer kept from an earlier call are not seen -- ask again after writing. */
C2HIP_CAPI void* c2hip_blob_data_ptr(c2hip_workspace* ws, const char* name);
/* Wrap externally owned device memory (e.g. a torch tensor) as a HIP tensor
 * blob without copying (Tensor::ShareExternalPointer, tensor.h). */
C2HIP_CAPI int c2hip_share_external(c2hip_workspace* ws, const char* name, void* device_ptr,
                                    const int64_t* dims, int ndim, int dtype, int device_id);

/* --- operators (pybind_state.cc run_operator_once; caffe2/core/
 *     operator.cc:116-200 CreateOperator) ---------------------------------
 * def_bytes: protobuf-serialized OperatorDef. */
C2HIP_CAPI int c2hip_run_operator_once(c2hip_workspace* ws, const void* def_bytes, size_t n);
C2HIP_CAPI c2hip_operator* c2hip_create_operator(c2hip_workspace* ws, const void* def_bytes,
                                                 size_t n);
/* sync != 0: reference semantics, the stream is synchronised and checked
 * after the op (operator.h:378); sync == 0: enqueue only. */
C2HIP_CAPI int c2hip_run_operator(c2hip_operator* op, int sync);
C2HIP_CAPI void c2hip_destroy_operator(c2hip_operator* op);

/* --- nets (caffe2/python/workspace.py CreateNet / RunNet / RunNetOnce -> pybind_state.cc
 *     "create_net" / "run_net" / "run_net_once" -> caffe2/core/workspace.cc:180-260; the training loop is
 *     one workspace.RunNet(model.net.Proto().name) per iteration, detectron/tools/train_net.py:165-189) ---
 * netdef_bytes: protobuf-serialized NetDef (caffe2.proto:176-215).  The net is instantiated ONCE: its
 * operator list is lowered for the MI355X kernels (Conv + Relu fusion, one multi-level launch per shared
 * filter, the shared-filter gradient Sum absorbed: csrc/c2/net.h, csrc/ops/net_lowering.cc; NetDef arg
 * hip_lowering = 0 keeps the list as written) and its operators are created; a run enqueues them in order
 * and synchronises once (NetDef arg hip_sync_every_op = 1: after every operator, as the reference's
 * executors do).  Results are those of running the operators one by one. */
C2HIP_CAPI int c2hip_create_net(c2hip_workspace* ws, const void* netdef_bytes, size_t n, int overwrite);
C2HIP_CAPI int c2hip_run_net(c2hip_workspace* ws, const char* name, int num_iter);
C2HIP_CAPI int c2hip_run_net_once(c2hip_workspace* ws, const void* netdef_bytes, size_t n);
C2HIP_CAPI int c2hip_delete_net(c2hip_workspace* ws, const char* name);
/* '\n'-separated net names; returns the byte count needed */
C2HIP_CAPI size_t c2hip_nets(c2hip_workspace* ws, char* buf, size_t buflen);
/* The operator list a created net actually runs, as [u32 length][OperatorDef bytes]...; *n_ops receives
 * the count.  Returns the byte count needed (0 + error if the net does not exist). */
C2HIP_CAPI size_t c2hip_net_lowered_ops(c2hip_workspace* ws, const char* name, void* buf, size_t buflen,
                                        int* n_ops);
/* The lowering alone, without a workspace or a device (host logic; filters are assumed fp32): writes the
 * lowered operator list as above and a one-line report into report_buf. */
C2HIP_CAPI size_t c2hip_lower_net(const void* netdef_bytes, size_t n, void* buf, size_t buflen, int* n_ops,
                                  char* report_buf, size_t report_buflen);
/* Process-wide event counters for tests and benchmarks: "filter_packs" (3x3 filter packs issued by the
 * operators' pack cache), "conv_launch_calls" (3x3 launcher calls made by Conv / ConvGradient / the group
 * operators).  -1 for an unknown name. */
C2HIP_CAPI long long c2hip_counter(const char* name);

/* --- registry / schema / gradients (core.RefreshRegisteredOperators,
 *     core.GetGradientForOp -> pybind get_gradient_defs) ------------------- */
/* '\n'-separated registered keys for a device type; returns bytes needed */
C2HIP_CAPI size_t c2hip_registered_operators(int device_type, char* buf, size_t buflen);
C2HIP_CAPI int c2hip_has_schema(const char* op_type, int* min_in, int* max_in, int* min_out,
                                int* max_out);
/* g_output_names: '\n'-separated gradient blob names of the op's outputs
 * (empty entry = not provided).  Writes the gradient OperatorDefs as
 * [u32 length][bytes]... into out_defs and the '\n'-separated gradient names
 * of the op's inputs (empty = none) into out_g_inputs.  Returns 0 or error;
 * *n_defs receives the number of gradient operators. */
C2HIP_CAPI int c2hip_get_gradient_defs(const void* def_bytes, size_t n,
                                       const char* g_output_names, void* out_defs,
                                       size_t out_defs_cap, size_t* out_defs_len, int* n_defs,
                                       char* out_g_inputs, size_t out_g_inputs_cap);

/* --- stream plumbing -------------------------------------------------------
 * Operators of device_id enqueue on `hip_stream` (a hipStream_t) until
 * cleared with enabled = 0; default is an internal per-device stream
 * (caffe2/core/context_gpu.h:178-188 thread-local stream pool). */
C2HIP_CAPI int c2hip_set_stream(int device_id, void* hip_stream, int enabled);
C2HIP_CAPI int c2hip_device_synchronize(int device_id);

/* --- the data-parallel communicator of this process -------------------------
 * The reference's DP net carries its exchange as operators: model.net.NCCLAllreduce(gradients, gradients)
 * per parameter (detectron/lib/modeling/optimizer.py:72-92; caffe2/contrib/nccl/cuda_nccl_op_gpu.cc:25-118),
 * with one process owning every GPU (ncclCommInitAll, cuda_nccl_gpu.cc:141-175).  Here a replica is a
 * process: rank 0 creates an id (c2hip_comm_unique_id, 128 bytes = ncclUniqueId), the launcher hands it to
 * every rank, each calls c2hip_comm_init (ncclCommInitRank on RCCL over xGMI, loaded with dlopen), and from
 * then on the registered `NCCLAllreduce` / `NCCLBroadcast` operators -- one blob per rank, in place allowed --
 * run over that communicator on the operator's stream.  Without a communicator they are the reference's
 * single-GPU no-ops.  Errors: c2hip_comm_last_error(). */
C2HIP_CAPI const char* c2hip_comm_last_error(void);
C2HIP_CAPI int c2hip_comm_unique_id(void* id_out, size_t nbytes);
C2HIP_CAPI int c2hip_comm_init(const void* id, size_t nbytes, int world, int rank, int device_id);
C2HIP_CAPI int c2hip_comm_world(void);          /* 0 = no communicator */
C2HIP_CAPI int c2hip_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* C2HIP_CAPI_H_ */
