// hb_ops_shim.cc -- the ONLY FFI layer between TensorFlow and libhbk_core.so.
//
// Registers HybridBackend's custom ops for the sharded-embedding path under their existing
// names and signatures (so graphs produced by the reference's Python / graph passes load
// unchanged) and forwards each kernel to the C ABI of include/hbk.h.  No CUDA headers, no
// NCCL headers, no backend dispatch: everything device-side lives behind hbk_*.
//
// Not BUILT in this repository (TensorFlow is absent from the build image and the GPU box; build
// line in INTEGRATION.md), but parsed and type-checked by `make -C integration/tf_shim check`
// against a declaration stub of the TensorFlow symbols used here (check/tf_decl_stub.h).
// Signature sources, relative to the reference tree:
//   hybridbackend/tensorflow/distribute/partition/partition_by_modulo_ops.cc:46-60,124-143
//   hybridbackend/tensorflow/distribute/partition/partition_by_dual_modulo_ops.cc:46-61,132-147,184-204,278-298
//   hybridbackend/tensorflow/distribute/nccl/nccl_get_id.cc:35-41, nccl_create.cc:32-52
//   hybridbackend/tensorflow/distribute/nccl/nccl_alltoall.cc:169-180,242-258
//   hybridbackend/tensorflow/distribute/nccl/nccl_alltoallv.cc:200-223,359-387
//   hybridbackend/tensorflow/embedding/lookup_ops.cc:38-58
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hbk.h"
#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/register_types.h"
#include "tensorflow/core/framework/resource_mgr.h"
#include "tensorflow/core/framework/shape_inference.h"
#include "tensorflow/core/lib/core/threadpool.h"

namespace tensorflow {
namespace hybridbackend {

using shape_inference::InferenceContext;
using GPUDevice = Eigen::GpuDevice;

// ---- helpers -------------------------------------------------------------------------------
static Status HbkStatus(int rc) {  // hbk status codes are TensorFlow error codes
  if (rc == HBK_OK) return Status::OK();
  return Status(static_cast<error::Code>(rc), hbk_last_error());
}

static hbk_stream_t StreamOf(OpKernelContext* ctx) {
  return reinterpret_cast<hbk_stream_t>(ctx->eigen_device<GPUDevice>().stream());
}

template <typename T> struct HbkType;
#define HBK_TYPE(T, V) template <> struct HbkType<T> { static constexpr int32_t v = V; }
HBK_TYPE(int8, HBK_INT8); HBK_TYPE(uint8, HBK_UINT8); HBK_TYPE(int32, HBK_INT32);
HBK_TYPE(uint32, HBK_UINT32); HBK_TYPE(int64, HBK_INT64); HBK_TYPE(uint64, HBK_UINT64);
HBK_TYPE(Eigen::half, HBK_HALF); HBK_TYPE(float, HBK_FLOAT); HBK_TYPE(double, HBK_DOUBLE);
#undef HBK_TYPE

// TF_DETERMINISTIC_OPS=1 (TensorFlow's switch for bit-reproducible GPU kernels; the reference's
// gradient of the lookup is TF's UnsortedSegmentSum, which that switch makes deterministic): the
// backward's duplicate-row reduction sums every row's terms in id order -- bit-equal to TF's CPU
// kernel, rows ascending (hbk option bwd_deterministic = 1, include/hbk.h).  Read once, when the op
// library is loaded; HBK_BWD_DETERMINISTIC in the environment, if set, has already chosen.
static const bool kDeterministicOps = [] {
  const char* tf = std::getenv("TF_DETERMINISTIC_OPS");
  const bool on = tf != nullptr && (std::strcmp(tf, "1") == 0 || std::strcmp(tf, "true") == 0 ||
                                    std::strcmp(tf, "True") == 0);
  if (on && std::getenv("HBK_BWD_DETERMINISTIC") == nullptr) (void)hbk_set_option("bwd_deterministic", 1);
  return on;
}();

static Status AllocScratch(OpKernelContext* ctx, size_t bytes, Tensor* t) {
  return ctx->allocate_temp(DT_INT8, TensorShape({static_cast<int64>(bytes) + 16}), t);
}

// ============================================================================================
// HbPartitionByModulo[N], HbPartitionByDualModuloStage{One,Two}[N]
// ============================================================================================
#define HB_PARTITION_SHAPE_FN(N_EXPR)                                              \
  [](InferenceContext* c) {                                                        \
    int64 n = (N_EXPR);                                                            \
    int32 num_partitions;                                                          \
    TF_RETURN_IF_ERROR(c->GetAttr("num_partitions", &num_partitions));             \
    for (int64 i = 0; i < n; ++i) {                                                \
      c->set_output(i, c->input(i));                                               \
      c->set_output(n + i, c->Vector(num_partitions));                             \
      c->set_output(2 * n + i, c->input(i));                                       \
    }                                                                              \
    return Status::OK();                                                           \
  }

REGISTER_OP("HbPartitionByModulo")
    .Output("output: T").Output("sizes: int32").Output("indices: int32")
    .Input("input: T")
    .Attr("T: {int32, int64, uint32, uint64}")
    .Attr("num_partitions: int >= 1 = 1")
    .SetShapeFn(HB_PARTITION_SHAPE_FN(1));

REGISTER_OP("HbPartitionByModuloN")
    .Output("outputs: N * T").Output("outputs_sizes: N * int32")
    .Output("outputs_indices: N * int32")
    .Input("inputs: N * T")
    .Attr("N: int >= 1 = 1")
    .Attr("T: {int32, int64, uint32, uint64}")
    .Attr("num_partitions: int >= 1 = 1")
    .SetShapeFn([](InferenceContext* c) {
      int64 n;
      TF_RETURN_IF_ERROR(c->GetAttr("N", &n));
      int32 p;
      TF_RETURN_IF_ERROR(c->GetAttr("num_partitions", &p));
      for (int64 i = 0; i < n; ++i) {
        c->set_output(i, c->input(i));
        c->set_output(n + i, c->Vector(p));
        c->set_output(2 * n + i, c->input(i));
      }
      return Status::OK();
    });

#define HB_REGISTER_DUAL_OPS(STAGE)                                                          \
  REGISTER_OP("HbPartitionByDualModuloStage" #STAGE)                                         \
      .Output("output: T").Output("sizes: int32").Output("indices: int32")                  \
      .Input("input: T")                                                                     \
      .Attr("T: {int32, int64, uint32, uint64}")                                             \
      .Attr("num_partitions: int >= 1 = 1").Attr("modulus: int >= 1 = 1")                   \
      .SetShapeFn(HB_PARTITION_SHAPE_FN(1));                                                 \
  REGISTER_OP("HbPartitionByDualModuloStage" #STAGE "N")                                     \
      .Output("outputs: N * T").Output("outputs_sizes: N * int32")                          \
      .Output("outputs_indices: N * int32")                                                  \
      .Input("inputs: N * T")                                                                \
      .Attr("N: int >= 1 = 1").Attr("T: {int32, int64, uint32, uint64}")                    \
      .Attr("num_partitions: int >= 1 = 1").Attr("modulus: int >= 1 = 1")
HB_REGISTER_DUAL_OPS(One);
HB_REGISTER_DUAL_OPS(Two);

// One kernel class for all six partition ops: `stage` 0 = plain modulo, 1 / 2 = dual modulo;
// `nary` selects list inputs.  Everything is enqueued on the op's compute stream.
template <typename T>
class PartitionOp : public OpKernel {
 public:
  PartitionOp(OpKernelConstruction* ctx, int stage, bool nary)
      : OpKernel(ctx), stage_(stage), nary_(nary), modulus_(1) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("num_partitions", &num_partitions_));
    if (stage_ != 0) OP_REQUIRES_OK(ctx, ctx->GetAttr("modulus", &modulus_));
  }

  void Compute(OpKernelContext* ctx) override {
    std::vector<const Tensor*> in;
    if (nary_) {
      OpInputList list;
      OP_REQUIRES_OK(ctx, ctx->input_list("inputs", &list));
      for (int i = 0; i < list.size(); ++i) in.push_back(&list[i]);
    } else {
      in.push_back(&ctx->input(0));
    }
    const int n = static_cast<int>(in.size());
    std::vector<const void*> src(n);
    std::vector<void*> dst(n);
    std::vector<int32_t*> sizes(n), idx(n);
    std::vector<int64_t> lens(n);
    for (int i = 0; i < n; ++i) {
      OP_REQUIRES(ctx, TensorShapeUtils::IsVector(in[i]->shape()),
                  errors::InvalidArgument("Input must be a vector"));
      Tensor *o, *s, *x;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, in[i]->shape(), &o));
      OP_REQUIRES_OK(ctx, ctx->allocate_output(n + i, TensorShape({num_partitions_}), &s));
      OP_REQUIRES_OK(ctx, ctx->allocate_output(2 * n + i, in[i]->shape(), &x));
      src[i] = in[i]->flat<T>().data();
      dst[i] = o->flat<T>().data();
      sizes[i] = s->flat<int32>().data();
      idx[i] = x->flat<int32>().data();
      lens[i] = in[i]->NumElements();
    }
    const size_t ws_bytes = hbk_partition_workspace_bytes(n, lens.data(), num_partitions_);
    Tensor ws;
    OP_REQUIRES_OK(ctx, AllocScratch(ctx, ws_bytes, &ws));
    void* wsp = ws.flat<int8>().data();
    int rc;
    if (stage_ == 0) {
      rc = hbk_partition_by_modulo_n(n, HbkType<T>::v, num_partitions_, src.data(),
                                     lens.data(), dst.data(), sizes.data(), idx.data(), wsp,
                                     ws_bytes + 16, StreamOf(ctx));
    } else {
      rc = hbk_partition_by_dual_modulo_n(n, HbkType<T>::v, num_partitions_, modulus_, stage_,
                                          src.data(), lens.data(), dst.data(), sizes.data(),
                                          idx.data(), wsp, ws_bytes + 16, StreamOf(ctx));
    }
    OP_REQUIRES_OK(ctx, HbkStatus(rc));
  }

 private:
  int stage_;
  bool nary_;
  int32 num_partitions_;
  int32 modulus_;
};

#define HB_PARTITION_KERNEL(NAME, STAGE, NARY, T)                                        \
  class NAME##Kernel##T : public PartitionOp<T> {                                        \
   public:                                                                               \
    explicit NAME##Kernel##T(OpKernelConstruction* c) : PartitionOp<T>(c, STAGE, NARY) {} \
  };                                                                                     \
  REGISTER_KERNEL_BUILDER(Name(#NAME).Device(DEVICE_GPU).TypeConstraint<T>("T"),         \
                          NAME##Kernel##T)
#define HB_PARTITION_KERNELS(T)                                   \
  HB_PARTITION_KERNEL(HbPartitionByModulo, 0, false, T);           \
  HB_PARTITION_KERNEL(HbPartitionByModuloN, 0, true, T);           \
  HB_PARTITION_KERNEL(HbPartitionByDualModuloStageOne, 1, false, T); \
  HB_PARTITION_KERNEL(HbPartitionByDualModuloStageOneN, 1, true, T); \
  HB_PARTITION_KERNEL(HbPartitionByDualModuloStageTwo, 2, false, T); \
  HB_PARTITION_KERNEL(HbPartitionByDualModuloStageTwoN, 2, true, T)
HB_PARTITION_KERNELS(int32);
HB_PARTITION_KERNELS(int64);
HB_PARTITION_KERNELS(uint32);
HB_PARTITION_KERNELS(uint64);

// CPU kernels of the non-N ops (the reference registers them for DEVICE_CPU:
// partition_by_modulo_ops.cc:62-101, partition_by_dual_modulo_ops.cc:62-130): host tensors
// through the host-memory twins of the device entries.
template <typename T>
class PartitionCpuOp : public OpKernel {
 public:
  PartitionCpuOp(OpKernelConstruction* ctx, int stage) : OpKernel(ctx), stage_(stage), modulus_(1) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("num_partitions", &num_partitions_));
    if (stage_ != 0) OP_REQUIRES_OK(ctx, ctx->GetAttr("modulus", &modulus_));
  }
  void Compute(OpKernelContext* ctx) override {
    const Tensor& in = ctx->input(0);
    OP_REQUIRES(ctx, TensorShapeUtils::IsVector(in.shape()),
                errors::InvalidArgument("Input must be a vector"));
    Tensor *o, *s, *x;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, in.shape(), &o));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, TensorShape({num_partitions_}), &s));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(2, in.shape(), &x));
    const void* src = in.flat<T>().data();
    void* dst = o->flat<T>().data();
    int32_t* sizes = s->flat<int32>().data();
    int32_t* idx = x->flat<int32>().data();
    const int64_t len = in.NumElements();
    const int rc = stage_ == 0
        ? hbk_partition_by_modulo_host(1, HbkType<T>::v, num_partitions_, &src, &len, &dst,
                                       &sizes, &idx)
        : hbk_partition_by_dual_modulo_host(1, HbkType<T>::v, num_partitions_, modulus_, stage_,
                                            &src, &len, &dst, &sizes, &idx);
    OP_REQUIRES_OK(ctx, HbkStatus(rc));
  }

 private:
  int stage_;
  int32 num_partitions_;
  int32 modulus_;
};
#define HB_PARTITION_CPU_KERNEL(NAME, STAGE, T)                                          \
  class NAME##CpuKernel##T : public PartitionCpuOp<T> {                                  \
   public:                                                                               \
    explicit NAME##CpuKernel##T(OpKernelConstruction* c) : PartitionCpuOp<T>(c, STAGE) {} \
  };                                                                                     \
  REGISTER_KERNEL_BUILDER(Name(#NAME).Device(DEVICE_CPU).TypeConstraint<T>("T"),         \
                          NAME##CpuKernel##T)
#define HB_PARTITION_CPU_KERNELS(T)                                  \
  HB_PARTITION_CPU_KERNEL(HbPartitionByModulo, 0, T);                 \
  HB_PARTITION_CPU_KERNEL(HbPartitionByDualModuloStageOne, 1, T);     \
  HB_PARTITION_CPU_KERNEL(HbPartitionByDualModuloStageTwo, 2, T)
HB_PARTITION_CPU_KERNELS(int32);
HB_PARTITION_CPU_KERNELS(int64);
HB_PARTITION_CPU_KERNELS(uint32);
HB_PARTITION_CPU_KERNELS(uint64);

// ============================================================================================
// Communicator resource: HbGetNcclId / HbNcclCollectiveHandleOp / HbCreateNcclCollective /
// HbIsNcclCollectiveInitialized.  The resource owns an hbk_comm_t (RCCL communicator + its
// private stream) and a small thread pool for the async ops, like NcclCollective does.
// ============================================================================================
class HbNcclCollective : public ResourceBase {
 public:
  HbNcclCollective() : comm_(nullptr), pool_(nullptr) {}
  ~HbNcclCollective() override {
    delete pool_;
    if (comm_ != nullptr) hbk_comm_destroy(comm_);
  }
  Status Create(const uint8_t* id, int world, int local, int rank) {
    TF_RETURN_IF_ERROR(HbkStatus(hbk_comm_create(&comm_, id, world, local, rank)));
    pool_ = new thread::ThreadPool(Env::Default(), "hbk_collective", 3);
    return Status::OK();
  }
  bool initialized() const { return comm_ != nullptr; }
  hbk_comm_t comm() const { return comm_; }
  thread::ThreadPool* pool() const { return pool_; }
  string DebugString() const override { return "HbNcclCollective(libhbk_core)"; }

 private:
  hbk_comm_t comm_;
  thread::ThreadPool* pool_;
};

REGISTER_RESOURCE_HANDLE_OP(HbNcclCollective);
REGISTER_KERNEL_BUILDER(Name("HbNcclCollectiveHandleOp").Device(DEVICE_GPU),
                        ResourceHandleOp<HbNcclCollective>);

REGISTER_OP("HbGetNcclId")
    .Output("id: int64")
    .SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      c->set_output(0, c->Vector(HBK_COMM_ID_BYTES / sizeof(int64)));
      return Status::OK();
    });

class GetNcclIdOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    Tensor* id;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(
                            0, TensorShape({HBK_COMM_ID_BYTES / sizeof(int64)}), &id));
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_comm_get_id(
                            reinterpret_cast<uint8_t*>(id->flat<int64>().data()))));
  }
};
REGISTER_KERNEL_BUILDER(Name("HbGetNcclId").Device(DEVICE_GPU).HostMemory("id"), GetNcclIdOp);
REGISTER_KERNEL_BUILDER(Name("HbGetNcclId").Device(DEVICE_CPU), GetNcclIdOp);

REGISTER_OP("HbCreateNcclCollective")
    .Input("handle: resource").Input("id: int64")
    .Attr("world_size: int").Attr("local_size: int").Attr("rank: int")
    .Attr("shared_name: string")
    .SetShapeFn(shape_inference::NoOutputs);

class CreateNcclCollectiveOp : public OpKernel {
 public:
  explicit CreateNcclCollectiveOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("world_size", &world_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("local_size", &local_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("rank", &rank_));
  }
  void Compute(OpKernelContext* ctx) override {
    const Tensor& id = ctx->input(1);
    OP_REQUIRES(ctx, id.NumElements() * sizeof(int64) == HBK_COMM_ID_BYTES,
                errors::InvalidArgument("id must hold ", HBK_COMM_ID_BYTES, " bytes"));
    HbNcclCollective* coll = new HbNcclCollective();
    Status s = coll->Create(reinterpret_cast<const uint8_t*>(id.flat<int64>().data()), world_,
                            local_, rank_);
    if (!s.ok()) {
      coll->Unref();
      OP_REQUIRES_OK(ctx, s);
    }
    OP_REQUIRES_OK(ctx, CreateResource(ctx, HandleFromInput(ctx, 0), coll));
  }

 private:
  int world_, local_, rank_;
};
REGISTER_KERNEL_BUILDER(Name("HbCreateNcclCollective").Device(DEVICE_GPU).HostMemory("id"),
                        CreateNcclCollectiveOp);

REGISTER_OP("HbIsNcclCollectiveInitialized")
    .Output("is_initialized: bool").Input("handle: resource")
    .SetShapeFn(shape_inference::ScalarShape);

class IsNcclCollectiveInitializedOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    Tensor* out;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({}), &out));
    HbNcclCollective* coll = nullptr;
    const bool found = LookupResource(ctx, HandleFromInput(ctx, 0), &coll).ok();
    out->scalar<bool>()() = found && coll->initialized();
    if (found) coll->Unref();
  }
};
REGISTER_KERNEL_BUILDER(
    Name("HbIsNcclCollectiveInitialized").Device(DEVICE_GPU).HostMemory("is_initialized"),
    IsNcclCollectiveInitializedOp);

// ============================================================================================
// HbNcclAlltoall[N] (equal split) and HbNcclAlltoallv[N]
// ============================================================================================
#define HB_DTYPES "{int8, uint8, int32, uint32, int64, uint64, half, float, double}"
#define HB_WIRE_DTYPES "{half, float}"

// Shape functions as the reference declares them (nccl_alltoall.cc:169-180,242-258,
// nccl_alltoallv.cc:200-223,359-387): the equal-split ops keep their input's shape, the
// Alltoallv ops give [?, common_shape...] and pass input_sizes' shape through -- graph
// construction downstream of the exchange relies on these static shapes.
static Status AlltoallvShape(InferenceContext* c, const PartialTensorShape& common, int out,
                             int sizes_out, int sizes_in) {
  shape_inference::ShapeHandle tail, full;
  TF_RETURN_IF_ERROR(c->MakeShapeFromPartialTensorShape(common, &tail));
  TF_RETURN_IF_ERROR(c->Concatenate(c->Vector(InferenceContext::kUnknownDim), tail, &full));
  c->set_output(out, full);
  c->set_output(sizes_out, c->input(sizes_in));
  return Status::OK();
}

REGISTER_OP("HbNcclAlltoall")
    .Output("output: dtype").Input("handle: resource").Input("input: dtype")
    .Attr("topology: int = 0").Attr("dtype: " HB_DTYPES).Attr("wire_dtype: " HB_WIRE_DTYPES)
    .SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      c->set_output(0, c->input(1));   // (input 0 is the communicator handle)
      return Status::OK();
    });
REGISTER_OP("HbNcclAlltoallN")
    .Output("n_output: N * dtype").Input("handle: resource").Input("n_input: N * dtype")
    .Attr("N: int >= 1 = 1").Attr("topology: int = 0").Attr("dtype: " HB_DTYPES)
    .Attr("wire_dtype: " HB_WIRE_DTYPES).SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      int64 n;
      TF_RETURN_IF_ERROR(c->GetAttr("N", &n));
      for (int64 i = 0; i < n; ++i) c->set_output(i, c->input(1 + i));
      return Status::OK();
    });
REGISTER_OP("HbNcclAlltoallv")
    .Output("output: dtype").Output("output_sizes: int32")
    .Input("handle: resource").Input("input: dtype").Input("input_sizes: int32")
    .Attr("common_shape: shape = {}").Attr("topology: int = 0").Attr("dtype: " HB_DTYPES)
    .Attr("wire_dtype: " HB_WIRE_DTYPES).SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      PartialTensorShape common;
      TF_RETURN_IF_ERROR(c->GetAttr("common_shape", &common));
      return AlltoallvShape(c, common, 0, 1, 2);
    });
REGISTER_OP("HbNcclAlltoallvN")
    .Output("n_output: N * dtype").Output("n_output_sizes: N * int32")
    .Input("handle: resource").Input("n_input: N * dtype").Input("n_input_sizes: N * int32")
    .Attr("N: int >= 1 = 1").Attr("common_shape: list(shape)").Attr("topology: int = 0")
    .Attr("dtype: " HB_DTYPES).Attr("wire_dtype: " HB_WIRE_DTYPES).SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      int64 n;
      TF_RETURN_IF_ERROR(c->GetAttr("N", &n));
      std::vector<PartialTensorShape> common;
      TF_RETURN_IF_ERROR(c->GetAttr("common_shape", &common));
      for (int64 i = 0; i < n; ++i) {
        TF_RETURN_IF_ERROR(AlltoallvShape(c, common[i], i, n + i, 1 + n + i));
      }
      return Status::OK();
    });

// Base of the collective ops: looks the communicator up and hops to its thread pool, because
// the Alltoallv ops block the calling thread once (sizes must reach the host before the
// outputs can be allocated -- the reference does the same, nccl_alltoallv.cc:306-329).
class CollectiveAsyncOp : public AsyncOpKernel {
 public:
  using AsyncOpKernel::AsyncOpKernel;
  void ComputeAsync(OpKernelContext* ctx, DoneCallback done) override {
    HbNcclCollective* coll = nullptr;
    OP_REQUIRES_OK_ASYNC(ctx, LookupResource(ctx, HandleFromInput(ctx, 0), &coll), done);
    // an asynchronous RCCL error aborts the communicator and fails the step here instead of
    // hanging it (the reference polls ncclCommGetAsyncError from a thread of its own,
    // nccl_collective.cc:434-465)
    const Status healthy = HbkStatus(hbk_comm_check_async(coll->comm()));
    if (!healthy.ok()) coll->Unref();
    OP_REQUIRES_OK_ASYNC(ctx, healthy, done);
    coll->pool()->Schedule([this, ctx, coll, done]() {
      Run(ctx, coll);
      coll->Unref();
      done();
    });
  }
  virtual void Run(OpKernelContext* ctx, HbNcclCollective* coll) = 0;
};

template <typename T>
class AlltoallNOp : public CollectiveAsyncOp {
 public:
  explicit AlltoallNOp(OpKernelConstruction* ctx) : CollectiveAsyncOp(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("topology", &topology_));
  }
  void Run(OpKernelContext* ctx, HbNcclCollective* coll) override {
    const int n = ctx->num_inputs() - 1;
    std::vector<const void*> in(n);
    std::vector<void*> out(n);
    std::vector<int64_t> counts(n);
    for (int i = 0; i < n; ++i) {
      const Tensor& t = ctx->input(1 + i);
      Tensor* o;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, t.shape(), &o));
      in[i] = t.tensor_data().data();
      out[i] = const_cast<char*>(o->tensor_data().data());
      counts[i] = t.NumElements();
    }
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_alltoall_n(coll->comm(), n, HbkType<T>::v, topology_,
                                                 in.data(), counts.data(), out.data(),
                                                 StreamOf(ctx))));
  }

 private:
  int topology_;
};

template <typename T, typename W>
class AlltoallvNOp : public CollectiveAsyncOp {
 public:
  explicit AlltoallvNOp(OpKernelConstruction* ctx) : CollectiveAsyncOp(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("topology", &topology_));
    if (ctx->HasAttr("N")) {
      std::vector<PartialTensorShape> shapes;
      OP_REQUIRES_OK(ctx, ctx->GetAttr("common_shape", &shapes));
      common_shapes_ = shapes;
    } else {
      PartialTensorShape shape;
      OP_REQUIRES_OK(ctx, ctx->GetAttr("common_shape", &shape));
      common_shapes_.push_back(shape);
    }
  }
  void Run(OpKernelContext* ctx, HbNcclCollective* coll) override {
    const int n = static_cast<int>(common_shapes_.size());
    const int active = hbk_comm_active_ranks(coll->comm(), topology_, nullptr);
    // 1 exchange the sizes of all N tensors in one equal-split alltoall
    std::vector<const void*> sin(n);
    std::vector<void*> sout(n);
    std::vector<int64_t> scount(n, active);
    for (int i = 0; i < n; ++i) {
      const Tensor& sizes = ctx->input(1 + n + i);
      OP_REQUIRES(ctx, sizes.NumElements() == active,
                  errors::InvalidArgument("Sizes of input ", i, " must have ", active,
                                          " elements"));
      Tensor* osz;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(n + i, sizes.shape(), &osz));
      sin[i] = sizes.flat<int32>().data();
      sout[i] = osz->flat<int32>().data();
    }
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_alltoall_n(coll->comm(), n, HBK_INT32, topology_,
                                                 sin.data(), scount.data(), sout.data(),
                                                 StreamOf(ctx))));
    // 2 sizes to the host (the one host sync of the op)
    std::vector<int32_t> send(static_cast<size_t>(n) * active), recv(send.size());
    auto* stream = ctx->op_device_context()->stream();
    for (int i = 0; i < n; ++i) {
      se::DeviceMemoryBase s(const_cast<void*>(sin[i]), active * sizeof(int32));
      se::DeviceMemoryBase r(sout[i], active * sizeof(int32));
      stream->ThenMemcpy(&send[static_cast<size_t>(i) * active], s, active * sizeof(int32));
      stream->ThenMemcpy(&recv[static_cast<size_t>(i) * active], r, active * sizeof(int32));
    }
    OP_REQUIRES_OK(ctx, stream->BlockHostUntilDone());
    // 3 allocate outputs, 4 exchange the payload
    std::vector<const void*> in(n);
    std::vector<void*> out(n);
    std::vector<int64_t> common(n);
    for (int i = 0; i < n; ++i) {
      int64 rows = 0;
      for (int a = 0; a < active; ++a) rows += recv[static_cast<size_t>(i) * active + a];
      TensorShape shape;
      PartialTensorShape({rows}).Concatenate(common_shapes_[i]).AsTensorShape(&shape);
      Tensor* o;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, shape, &o));
      common[i] = 1;
      for (int d = 0; d < common_shapes_[i].dims(); ++d) common[i] *= common_shapes_[i].dim_size(d);
      in[i] = ctx->input(1 + i).tensor_data().data();
      out[i] = const_cast<char*>(o->tensor_data().data());
    }
    const size_t ws_bytes = hbk_alltoallv_wire_workspace_bytes(n, common.data(), send.data(),
                                                               recv.data(), active);
    Tensor ws;
    OP_REQUIRES_OK(ctx, AllocScratch(ctx, ws_bytes, &ws));
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_alltoallv_n(
                            coll->comm(), n, HbkType<T>::v, HbkType<W>::v, topology_,
                            common.data(), in.data(), send.data(), out.data(), recv.data(),
                            ws.flat<int8>().data(), ws_bytes + 16, StreamOf(ctx))));
  }

 private:
  int topology_;
  std::vector<PartialTensorShape> common_shapes_;
};

#define HB_REGISTER_COLLECTIVES(T)                                                          \
  REGISTER_KERNEL_BUILDER(Name("HbNcclAlltoall").Device(DEVICE_GPU).TypeConstraint<T>("dtype"), \
                          AlltoallNOp<T>);                                                  \
  REGISTER_KERNEL_BUILDER(Name("HbNcclAlltoallN").Device(DEVICE_GPU).TypeConstraint<T>("dtype"), \
                          AlltoallNOp<T>);                                                  \
  REGISTER_KERNEL_BUILDER(Name("HbNcclAlltoallv").Device(DEVICE_GPU)                        \
                              .TypeConstraint<T>("dtype").TypeConstraint<float>("wire_dtype"), \
                          AlltoallvNOp<T, T>);                                              \
  REGISTER_KERNEL_BUILDER(Name("HbNcclAlltoallvN").Device(DEVICE_GPU)                       \
                              .TypeConstraint<T>("dtype").TypeConstraint<float>("wire_dtype"), \
                          AlltoallvNOp<T, T>)
HB_REGISTER_COLLECTIVES(int8); HB_REGISTER_COLLECTIVES(uint8); HB_REGISTER_COLLECTIVES(int32);
HB_REGISTER_COLLECTIVES(uint32); HB_REGISTER_COLLECTIVES(int64); HB_REGISTER_COLLECTIVES(uint64);
HB_REGISTER_COLLECTIVES(float); HB_REGISTER_COLLECTIVES(double);
// fp16 on the wire for fp32 data (comm_wire_dtype = half, collective.py:291-296)
REGISTER_KERNEL_BUILDER(Name("HbNcclAlltoallv").Device(DEVICE_GPU)
                            .TypeConstraint<float>("dtype").TypeConstraint<Eigen::half>("wire_dtype"),
                        AlltoallvNOp<float, Eigen::half>);
REGISTER_KERNEL_BUILDER(Name("HbNcclAlltoallvN").Device(DEVICE_GPU)
                            .TypeConstraint<float>("dtype").TypeConstraint<Eigen::half>("wire_dtype"),
                        AlltoallvNOp<float, Eigen::half>);

// ============================================================================================
// HbNcclAllreduce / HbNcclAllreduceN / HbNcclAllreduceMergedN, HbNcclAllgatherv
// (nccl_allreduce.cc:31-49,93-116,180-203; nccl_allgatherv.cc:31-60): gradient aggregation.
// All three allreduce forms go through hbk_allreduce_n, which always buckets its N tensors.
// ============================================================================================
#define HB_REGISTER_ALLREDUCE_OP(NAME, IN, OUT, EXTRA)                                        \
  REGISTER_OP(NAME).Output(OUT).Input("handle: resource").Input(IN)                           \
      .Attr("reduce_op: int >= 0 = 0").Attr("dtype: " HB_DTYPES) EXTRA.SetIsStateful()
HB_REGISTER_ALLREDUCE_OP("HbNcclAllreduce", "input: dtype", "output: dtype", );
HB_REGISTER_ALLREDUCE_OP("HbNcclAllreduceN", "n_input: N * dtype", "n_output: N * dtype",
                         .Attr("N: int >= 1 = 1"));
HB_REGISTER_ALLREDUCE_OP("HbNcclAllreduceMergedN", "n_input: N * dtype", "n_output: N * dtype",
                         .Attr("N: int >= 1 = 1"));

template <typename T>
class AllreduceNOp : public CollectiveAsyncOp {
 public:
  explicit AllreduceNOp(OpKernelConstruction* ctx) : CollectiveAsyncOp(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("reduce_op", &reduce_op_));
  }
  void Run(OpKernelContext* ctx, HbNcclCollective* coll) override {
    const int n = ctx->num_inputs() - 1;
    std::vector<const void*> in(n);
    std::vector<void*> out(n);
    std::vector<int64_t> counts(n);
    for (int i = 0; i < n; ++i) {
      const Tensor& t = ctx->input(1 + i);
      Tensor* o;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, t.shape(), &o));
      in[i] = t.tensor_data().data();
      out[i] = const_cast<char*>(o->tensor_data().data());
      counts[i] = t.NumElements();
    }
    const size_t ws_bytes = hbk_allreduce_workspace_bytes(n, counts.data(), HbkType<T>::v);
    Tensor ws;
    OP_REQUIRES_OK(ctx, AllocScratch(ctx, ws_bytes, &ws));
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_allreduce_n(
                            coll->comm(), n, HbkType<T>::v, reduce_op_, in.data(), counts.data(),
                            out.data(), 1.0f, ws.flat<int8>().data(), ws_bytes + 16,
                            StreamOf(ctx))));
  }

 private:
  int reduce_op_;
};

REGISTER_OP("HbNcclAllgatherv")
    .Output("output: dtype").Input("handle: resource").Input("input: dtype")
    .Attr("dtype: " HB_DTYPES).SetIsStateful();
// HbNcclAllgather (nccl_allgather.cc:31-101): every rank contributes the same shape, the output is
// [world x dim0, ...] in rank order.  HbNcclBroadcast (nccl_broadcast.cc:31-92): the root's tensor
// on every rank.
REGISTER_OP("HbNcclAllgather")
    .Output("output: dtype").Input("handle: resource").Input("input: dtype")
    .Attr("dtype: " HB_DTYPES).SetIsStateful();
REGISTER_OP("HbNcclBroadcast")
    .Output("output: dtype").Input("handle: resource").Input("input: dtype")
    .Attr("root_rank: int >= 0 = 0").Attr("dtype: " HB_DTYPES).SetIsStateful()
    .SetShapeFn([](shape_inference::InferenceContext* c) {
      c->set_output(0, c->input(1));
      return Status::OK();
    });

// Allgatherv: one hbk_alltoall_n of the local element count gives every rank's count, a host sync
// sizes the output (as nccl_allgatherv.cc:62-120 does), then hbk_allgatherv.  Allgather: the counts
// are known (every rank the same): no exchange, no sync.
template <typename T, bool EQUAL>
class AllgatherOp : public CollectiveAsyncOp {
 public:
  using CollectiveAsyncOp::CollectiveAsyncOp;
  void Run(OpKernelContext* ctx, HbNcclCollective* coll) override {
    const Tensor& in = ctx->input(1);
    const int W = hbk_comm_world_size(coll->comm());
    std::vector<int64_t> counts(W, in.NumElements());
    if (!EQUAL) {
      Tensor mine, all;
      OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_INT64, TensorShape({W}), &mine));
      OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_INT64, TensorShape({W}), &all));
      auto* stream = ctx->op_device_context()->stream();
      std::vector<int64> host_mine(W, in.NumElements());
      se::DeviceMemoryBase d_mine(mine.flat<int64>().data(), sizeof(int64) * W);
      stream->ThenMemcpy(&d_mine, host_mine.data(), sizeof(int64) * W);
      const void* vin[1] = {mine.flat<int64>().data()};
      void* vout[1] = {all.flat<int64>().data()};
      const int64_t cnt[1] = {W};
      OP_REQUIRES_OK(ctx, HbkStatus(hbk_alltoall_n(coll->comm(), 1, HBK_INT64, HBK_TOPOLOGY_ALL, vin,
                                                   cnt, vout, StreamOf(ctx))));
      std::vector<int64> host_all(W);
      se::DeviceMemoryBase d_all(all.flat<int64>().data(), sizeof(int64) * W);
      stream->ThenMemcpy(host_all.data(), d_all, sizeof(int64) * W);
      OP_REQUIRES_OK(ctx, stream->BlockHostUntilDone());
      for (int r = 0; r < W; ++r) counts[r] = host_all[r];
    }
    int64 total = 0;
    for (int r = 0; r < W; ++r) total += counts[r];
    // rows along dim 0; a scalar contributes one element
    int64 inner = 1;
    for (int d = 1; d < in.dims(); ++d) inner *= in.dim_size(d);
    TensorShape shape = in.dims() == 0 ? TensorShape({total}) : in.shape();
    if (in.dims() > 0) shape.set_dim(0, inner > 0 ? total / inner : 0);
    Tensor* out;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, shape, &out));
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_allgatherv(coll->comm(), HbkType<T>::v, in.tensor_data().data(),
                                                 counts.data(),
                                                 const_cast<char*>(out->tensor_data().data()),
                                                 StreamOf(ctx))));
  }
};

template <typename T>
class BroadcastOp : public CollectiveAsyncOp {
 public:
  explicit BroadcastOp(OpKernelConstruction* ctx) : CollectiveAsyncOp(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("root_rank", &root_));
  }
  void Run(OpKernelContext* ctx, HbNcclCollective* coll) override {
    const Tensor& in = ctx->input(1);
    Tensor* out;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, in.shape(), &out));
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_broadcast(coll->comm(), HbkType<T>::v, in.tensor_data().data(),
                                                const_cast<char*>(out->tensor_data().data()),
                                                in.NumElements(), root_, StreamOf(ctx))));
  }

 private:
  int root_;
};

#define HB_REGISTER_ALLREDUCE_KERNELS(T)                                                        \
  REGISTER_KERNEL_BUILDER(Name("HbNcclAllreduce").Device(DEVICE_GPU).TypeConstraint<T>("dtype"), \
                          AllreduceNOp<T>);                                                     \
  REGISTER_KERNEL_BUILDER(Name("HbNcclAllreduceN").Device(DEVICE_GPU).TypeConstraint<T>("dtype"), \
                          AllreduceNOp<T>);                                                     \
  REGISTER_KERNEL_BUILDER(                                                                      \
      Name("HbNcclAllreduceMergedN").Device(DEVICE_GPU).TypeConstraint<T>("dtype"),             \
      AllreduceNOp<T>)
HB_REGISTER_ALLREDUCE_KERNELS(int32); HB_REGISTER_ALLREDUCE_KERNELS(int64);
HB_REGISTER_ALLREDUCE_KERNELS(float); HB_REGISTER_ALLREDUCE_KERNELS(double);
HB_REGISTER_ALLREDUCE_KERNELS(Eigen::half);

#define HB_REGISTER_GATHER_KERNELS(T)                                                            \
  REGISTER_KERNEL_BUILDER(Name("HbNcclAllgatherv").Device(DEVICE_GPU).TypeConstraint<T>("dtype"), \
                          AllgatherOp<T, false>);                                                \
  REGISTER_KERNEL_BUILDER(Name("HbNcclAllgather").Device(DEVICE_GPU).TypeConstraint<T>("dtype"),  \
                          AllgatherOp<T, true>);                                                 \
  REGISTER_KERNEL_BUILDER(Name("HbNcclBroadcast").Device(DEVICE_GPU).TypeConstraint<T>("dtype"),  \
                          BroadcastOp<T>)
HB_REGISTER_GATHER_KERNELS(int8); HB_REGISTER_GATHER_KERNELS(uint8); HB_REGISTER_GATHER_KERNELS(int32);
HB_REGISTER_GATHER_KERNELS(uint32); HB_REGISTER_GATHER_KERNELS(int64); HB_REGISTER_GATHER_KERNELS(uint64);
HB_REGISTER_GATHER_KERNELS(float); HB_REGISTER_GATHER_KERNELS(double);
HB_REGISTER_GATHER_KERNELS(Eigen::half);

// ============================================================================================
// HbLookup (cache probe)
// ============================================================================================
REGISTER_OP("HbLookup")
    .Output("hit_keys_indices: Tindices").Output("hit_cache_indices: T")
    .Output("miss_keys_indices: Tindices").Output("miss_keys: T")
    .Input("keys_cache: T").Input("keys: T")
    .Attr("T: type").Attr("Tindices: {int32}").Attr("cache_slab_size: int");

// hbk_cache_lookup fills four capacity-n lists and {n_hit, n_miss} on the device; the op sizes
// its outputs from the counts after one host sync (as lookup_ops.cc:118-121 does) and copies the
// used prefixes out.
template <typename T>
class LookupOp : public OpKernel {
 public:
  explicit LookupOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("cache_slab_size", &slab_size_));
  }
  void Compute(OpKernelContext* ctx) override {
    const Tensor& cache = ctx->input(0);
    const Tensor& keys = ctx->input(1);
    const int64 n = keys.NumElements();
    OP_REQUIRES(ctx, slab_size_ >= 1 && cache.NumElements() % slab_size_ == 0,
                errors::InvalidArgument("keys_cache must hold a whole number of slabs"));
    Tensor hit_idx, hit_cache, miss_idx, miss_keys, counts, ws;
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_INT32, TensorShape({n}), &hit_idx));
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_INT64, TensorShape({n}), &hit_cache));
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_INT32, TensorShape({n}), &miss_idx));
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_INT64, TensorShape({n}), &miss_keys));
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_INT32, TensorShape({2}), &counts));
    const size_t ws_bytes = hbk_cache_lookup_workspace_bytes(n);
    OP_REQUIRES_OK(ctx, AllocScratch(ctx, ws_bytes, &ws));
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_cache_lookup(
        reinterpret_cast<const int64_t*>(cache.flat<T>().data()),
        cache.NumElements() / slab_size_, slab_size_,
        reinterpret_cast<const int64_t*>(keys.flat<T>().data()), n,
        hit_idx.flat<int32>().data(), reinterpret_cast<int64_t*>(hit_cache.flat<int64>().data()),
        miss_idx.flat<int32>().data(), reinterpret_cast<int64_t*>(miss_keys.flat<int64>().data()),
        counts.flat<int32>().data(), ws.flat<int8>().data(), ws_bytes + 16, StreamOf(ctx))));
    int32 host_counts[2];
    auto* stream = ctx->op_device_context()->stream();
    se::DeviceMemoryBase src(counts.flat<int32>().data(), sizeof(host_counts));
    stream->ThenMemcpy(host_counts, src, sizeof(host_counts));
    OP_REQUIRES_OK(ctx, stream->BlockHostUntilDone());
    const Tensor* parts[4] = {&hit_idx, &hit_cache, &miss_idx, &miss_keys};
    for (int i = 0; i < 4; ++i) {
      const int64 k = host_counts[i / 2];
      Tensor* out;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, TensorShape({k}), &out));
      if (k == 0) continue;
      const size_t bytes = k * (i % 2 == 0 ? sizeof(int32) : sizeof(int64));
      se::DeviceMemoryBase d(const_cast<char*>(out->tensor_data().data()), bytes);
      se::DeviceMemoryBase s(const_cast<char*>(parts[i]->tensor_data().data()), bytes);
      stream->ThenMemcpy(&d, s, bytes);
    }
  }

 private:
  int32 slab_size_;
};
REGISTER_KERNEL_BUILDER(
    Name("HbLookup").Device(DEVICE_GPU).TypeConstraint<int64>("T").TypeConstraint<int32>("Tindices"),
    LookupOp<int64>);

// ============================================================================================
// HbGroupLookup / HbGroupLookupGrad (new, additive; N-ary conventions of the Hb...N ops)
// ============================================================================================
REGISTER_OP("HbGroupLookup")
    .Output("outputs: N * float")
    .Input("weights: N * float").Input("ids: N * Tids").Input("row_splits: N * int32")
    .Attr("N: int >= 1").Attr("Tids: {int32, int64}")
    .Attr("buckets: list(int)").Attr("combiners: list(int)").Attr("ragged: list(bool)")
    .Attr("divisor: int = 1")
    // per column: skewed ids expected (Zipf heads) -- wide one-id-per-sample columns go through the
    // tiles that stage repeated rows in LDS (hbk_lookup_column_t.hot_rows); empty = none
    .Attr("hot_rows: list(bool) = []");

template <typename Tids>
class GroupLookupOp : public OpKernel {
 public:
  explicit GroupLookupOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("buckets", &buckets_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("combiners", &combiners_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("ragged", &ragged_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("divisor", &divisor_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("hot_rows", &hot_rows_));
  }
  void Compute(OpKernelContext* ctx) override {
    OpInputList w, ids, splits;
    OP_REQUIRES_OK(ctx, ctx->input_list("weights", &w));
    OP_REQUIRES_OK(ctx, ctx->input_list("ids", &ids));
    OP_REQUIRES_OK(ctx, ctx->input_list("row_splits", &splits));
    const int n = w.size();
    std::vector<hbk_lookup_column_t> cols(n);
    for (int i = 0; i < n; ++i) {
      const int64 n_seg = ragged_[i] ? splits[i].NumElements() - 1 : ids[i].NumElements();
      Tensor* o;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, TensorShape({n_seg, w[i].dim_size(1)}), &o));
      hbk_lookup_column_t& c = cols[i];
      std::memset(&c, 0, sizeof(c));
      c.table = w[i].flat<float>().data();
      c.rows = w[i].dim_size(0);
      c.dim = static_cast<int32_t>(w[i].dim_size(1));
      c.ids_dtype = HbkType<Tids>::v;
      c.ids = ids[i].flat<Tids>().data();
      c.n_ids = ids[i].NumElements();
      c.row_splits = ragged_[i] ? splits[i].flat<int32>().data() : nullptr;
      c.n_segments = n_seg;
      c.bucket = buckets_[i];
      c.divisor = divisor_;
      c.combiner = combiners_[i];
      c.hot_rows = i < static_cast<int>(hot_rows_.size()) && hot_rows_[i] ? 1 : 0;
      c.out = o->flat<float>().data();
    }
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_group_lookup_fwd(n, cols.data(), StreamOf(ctx))));
  }

 private:
  std::vector<int64> buckets_;
  std::vector<int32> combiners_;
  std::vector<bool> ragged_;
  std::vector<bool> hot_rows_;
  int32 divisor_;
};
REGISTER_KERNEL_BUILDER(Name("HbGroupLookup").Device(DEVICE_GPU).TypeConstraint<int32>("Tids"),
                        GroupLookupOp<int32>);
REGISTER_KERNEL_BUILDER(Name("HbGroupLookup").Device(DEVICE_GPU).TypeConstraint<int64>("Tids"),
                        GroupLookupOp<int64>);

// --------------------------------------------------------------------------------------------
// HbGroupLookupGrad: the gradient of HbGroupLookup with respect to `weights` as IndexedSlices
// (hbk_group_lookup_bwd): unique_rows / grad_rows have the capacity of the column's ids, the first
// n_unique[i][0] entries are valid; n_unique stays on the device.  The Python side ties it to
// HbGroupLookup with ops.RegisterGradient (INTEGRATION.md "Gradients"), the way the reference
// attaches its exchange gradients (hbtf/distribute/collective.py:334-347).
// HbGroupLookupGradApply: the same backward with the sparse optimizer step fused and NO
// IndexedSlices written ("step only": hbk_group_lookup_bwd_apply with unique_rows = grad_rows =
// NULL) -- sharded variables skip cross-rank aggregation (hbtf/training/gradient.py:193-217), so
// the step can be taken where the deduplicated sums sit in registers.
// --------------------------------------------------------------------------------------------
static Status GroupLookupGradShape(InferenceContext* c) {
  int64 n;
  TF_RETURN_IF_ERROR(c->GetAttr("N", &n));
  for (int64 i = 0; i < n; ++i) {      // inputs: weights [0,n), ids [n,2n), row_splits, grads
    c->set_output(i, c->Vector(c->Dim(c->input(n + i), 0)));
    c->set_output(n + i, c->Matrix(c->Dim(c->input(n + i), 0), c->Dim(c->input(i), 1)));
    c->set_output(2 * n + i, c->Vector(1));
  }
  return Status::OK();
}

REGISTER_OP("HbGroupLookupGrad")
    .Output("unique_rows: N * int64").Output("grad_rows: N * float").Output("n_unique: N * int32")
    .Input("weights: N * float").Input("ids: N * Tids").Input("row_splits: N * int32")
    .Input("grads: N * float")
    .Attr("N: int >= 1").Attr("Tids: {int32, int64}")
    .Attr("buckets: list(int)").Attr("combiners: list(int)").Attr("ragged: list(bool)")
    .Attr("divisor: int = 1")
    .Attr("deterministic: bool = false")   // sums in id order for this op (HBK_GRAD_DETERMINISTIC); TF_DETERMINISTIC_OPS=1 does it for all
    .SetShapeFn(GroupLookupGradShape);

REGISTER_OP("HbGroupLookupGradApply")
    .Output("n_unique: N * int32")
    .Input("weights: Ref(N * float)").Input("accums: Ref(M * float)")
    .Input("ids: N * Tids").Input("row_splits: N * int32").Input("grads: N * float")
    .Input("lr: float")
    .Attr("N: int >= 1").Attr("M: int >= 0 = 0").Attr("Tids: {int32, int64}")
    .Attr("buckets: list(int)").Attr("combiners: list(int)").Attr("ragged: list(bool)")
    .Attr("divisor: int = 1").Attr("optimizer: {'sgd', 'adagrad'} = 'sgd'")
    .Attr("deterministic: bool = false")
    .SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      int64 n;
      TF_RETURN_IF_ERROR(c->GetAttr("N", &n));
      for (int64 i = 0; i < n; ++i) c->set_output(i, c->Vector(1));
      return Status::OK();
    });

// the attributes both gradient ops share with HbGroupLookup, and the per-column descriptor
struct GroupLookupAttrs {
  std::vector<int64> buckets;
  std::vector<int32> combiners;
  std::vector<bool> ragged;
  int32 divisor;
  bool deterministic = false;
  Status Read(OpKernelConstruction* ctx) {
    TF_RETURN_IF_ERROR(ctx->GetAttr("buckets", &buckets));
    TF_RETURN_IF_ERROR(ctx->GetAttr("combiners", &combiners));
    TF_RETURN_IF_ERROR(ctx->GetAttr("ragged", &ragged));
    if (!ctx->GetAttr("deterministic", &deterministic).ok()) deterministic = false;   // (ops without the attr)
    return ctx->GetAttr("divisor", &divisor);
  }
  Status Check(int n) const {
    if (static_cast<int>(buckets.size()) != n || static_cast<int>(combiners.size()) != n ||
        static_cast<int>(ragged.size()) != n) {
      return errors::InvalidArgument("buckets, combiners and ragged must have N = ", n, " entries");
    }
    return Status::OK();
  }
  template <typename Tids>
  Status Fill(int i, const Tensor& weight, const Tensor& ids, const Tensor& splits,
              const Tensor& grad, hbk_lookup_grad_column_t* c) const {
    const int64 n_seg = ragged[i] ? splits.NumElements() - 1 : ids.NumElements();
    if (weight.dims() != 2 || grad.dims() != 2 || grad.dim_size(0) != n_seg ||
        grad.dim_size(1) != weight.dim_size(1)) {
      return errors::InvalidArgument("column ", i, ": grads must be [segments, dim] = [", n_seg,
                                     ", ", weight.dim_size(1), "]");
    }
    std::memset(c, 0, sizeof(*c));
    c->table = const_cast<float*>(weight.flat<float>().data());
    c->rows = weight.dim_size(0);
    c->dim = static_cast<int32_t>(weight.dim_size(1));
    c->ids_dtype = HbkType<Tids>::v;
    c->ids = ids.flat<Tids>().data();
    c->n_ids = ids.NumElements();
    c->row_splits = ragged[i] ? splits.flat<int32>().data() : nullptr;
    c->n_segments = n_seg;
    c->bucket = buckets[i];
    c->divisor = divisor;
    c->combiner = combiners[i];
    c->grad_out = grad.flat<float>().data();
    c->flags = deterministic ? HBK_GRAD_DETERMINISTIC : 0;
    return Status::OK();
  }
};

template <typename Tids>
class GroupLookupGradOp : public OpKernel {
 public:
  explicit GroupLookupGradOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, attrs_.Read(ctx));
  }
  void Compute(OpKernelContext* ctx) override {
    OpInputList w, ids, splits, grads;
    OP_REQUIRES_OK(ctx, ctx->input_list("weights", &w));
    OP_REQUIRES_OK(ctx, ctx->input_list("ids", &ids));
    OP_REQUIRES_OK(ctx, ctx->input_list("row_splits", &splits));
    OP_REQUIRES_OK(ctx, ctx->input_list("grads", &grads));
    const int n = w.size();
    OP_REQUIRES_OK(ctx, attrs_.Check(n));
    std::vector<hbk_lookup_grad_column_t> cols(n);
    for (int i = 0; i < n; ++i) {
      OP_REQUIRES_OK(ctx, attrs_.Fill<Tids>(i, w[i], ids[i], splits[i], grads[i], &cols[i]));
      Tensor *u, *g, *k;
      const int64 cap = ids[i].NumElements();
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, TensorShape({cap}), &u));
      OP_REQUIRES_OK(ctx, ctx->allocate_output(n + i, TensorShape({cap, w[i].dim_size(1)}), &g));
      OP_REQUIRES_OK(ctx, ctx->allocate_output(2 * n + i, TensorShape({1}), &k));
      cols[i].unique_rows = reinterpret_cast<int64_t*>(u->flat<int64>().data());
      cols[i].grad_rows = g->flat<float>().data();
      cols[i].n_unique = k->flat<int32>().data();
    }
    const size_t ws_bytes = hbk_group_lookup_bwd_workspace_bytes(n, cols.data());
    Tensor ws;
    OP_REQUIRES_OK(ctx, AllocScratch(ctx, ws_bytes, &ws));
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_group_lookup_bwd(n, cols.data(), 0.0f, ws.flat<int8>().data(),
                                                       ws_bytes + 16, StreamOf(ctx))));
  }

 private:
  GroupLookupAttrs attrs_;
};
REGISTER_KERNEL_BUILDER(Name("HbGroupLookupGrad").Device(DEVICE_GPU).TypeConstraint<int32>("Tids"),
                        GroupLookupGradOp<int32>);
REGISTER_KERNEL_BUILDER(Name("HbGroupLookupGrad").Device(DEVICE_GPU).TypeConstraint<int64>("Tids"),
                        GroupLookupGradOp<int64>);

static Status OptimizerCode(const string& name, int32_t* apply) {
  if (name == "sgd") { *apply = HBK_APPLY_SGD; return Status::OK(); }
  if (name == "adagrad") { *apply = HBK_APPLY_ADAGRAD; return Status::OK(); }
  return errors::InvalidArgument("optimizer must be 'sgd' or 'adagrad', got ", name);
}

template <typename Tids>
class GroupLookupGradApplyOp : public OpKernel {
 public:
  explicit GroupLookupGradApplyOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, attrs_.Read(ctx));
    string optimizer;
    OP_REQUIRES_OK(ctx, ctx->GetAttr("optimizer", &optimizer));
    OP_REQUIRES_OK(ctx, OptimizerCode(optimizer, &apply_));
  }
  void Compute(OpKernelContext* ctx) override {
    OpMutableInputList w, accums;
    OpInputList ids, splits, grads;
    OP_REQUIRES_OK(ctx, ctx->mutable_input_list("weights", &w));
    OP_REQUIRES_OK(ctx, ctx->mutable_input_list("accums", &accums));
    OP_REQUIRES_OK(ctx, ctx->input_list("ids", &ids));
    OP_REQUIRES_OK(ctx, ctx->input_list("row_splits", &splits));
    OP_REQUIRES_OK(ctx, ctx->input_list("grads", &grads));
    const Tensor* lr;
    OP_REQUIRES_OK(ctx, ctx->input("lr", &lr));
    const int n = w.size();
    OP_REQUIRES_OK(ctx, attrs_.Check(n));
    OP_REQUIRES(ctx, accums.size() == (apply_ == HBK_APPLY_ADAGRAD ? n : 0),
                errors::InvalidArgument("accums: N accumulators for adagrad, none for sgd"));
    std::vector<hbk_lookup_grad_column_t> cols(n);
    for (int i = 0; i < n; ++i) {
      Tensor weight = w.at(i, /*lock_held=*/false);
      OP_REQUIRES_OK(ctx, attrs_.Fill<Tids>(i, weight, ids[i], splits[i], grads[i], &cols[i]));
      if (apply_ == HBK_APPLY_ADAGRAD) {
        Tensor accum = accums.at(i, /*lock_held=*/false);
        OP_REQUIRES(ctx, accum.NumElements() == weight.NumElements(),
                    errors::InvalidArgument("accumulator ", i, " must have its variable's shape"));
        cols[i].accum = accum.flat<float>().data();
      }
      Tensor* k;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, TensorShape({1}), &k));
      cols[i].n_unique = k->flat<int32>().data();      // unique_rows = grad_rows = NULL: step only
    }
    const size_t ws_bytes = hbk_group_lookup_bwd_workspace_bytes(n, cols.data());
    Tensor ws;
    OP_REQUIRES_OK(ctx, AllocScratch(ctx, ws_bytes, &ws));
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_group_lookup_bwd_apply(
                            n, cols.data(), apply_, lr->scalar<float>()(), ws.flat<int8>().data(),
                            ws_bytes + 16, StreamOf(ctx))));
  }

 private:
  GroupLookupAttrs attrs_;
  int32_t apply_;
};
REGISTER_KERNEL_BUILDER(Name("HbGroupLookupGradApply").Device(DEVICE_GPU).HostMemory("lr")
                            .TypeConstraint<int32>("Tids"),
                        GroupLookupGradApplyOp<int32>);
REGISTER_KERNEL_BUILDER(Name("HbGroupLookupGradApply").Device(DEVICE_GPU).HostMemory("lr")
                            .TypeConstraint<int64>("Tids"),
                        GroupLookupGradApplyOp<int64>);

// ============================================================================================
// HbUniqueN / HbCastN (new, additive N-ary ops): the owner-side `array_ops.unique` of
// hbtf/embedding/sharding.py:186 for N columns in one set of launches (TensorFlow 1.15 has no GPU
// kernel for Unique: the stock op costs a device -> host -> device round trip per column), and the
// fp32 <-> fp16 wire casts of hbtf/common/cast.cu.cc:84-285 as a graph-level op (the exchange ops
// above cast inside hbk_alltoallv_n; this one serves graphs that keep fp16 rows around).
// ============================================================================================
REGISTER_OP("HbUniqueN")
    .Output("unique: N * int64").Output("index: N * int32").Output("n_unique: N * int32")
    .Input("inputs: N * int64").Attr("N: int >= 1 = 1")
    .SetShapeFn([](InferenceContext* c) {
      int64 n;
      TF_RETURN_IF_ERROR(c->GetAttr("N", &n));
      for (int64 i = 0; i < n; ++i) {     // capacity outputs: the first n_unique[i][0] ids are valid
        c->set_output(i, c->input(i));
        c->set_output(n + i, c->input(i));
        c->set_output(2 * n + i, c->Vector(1));
      }
      return Status::OK();
    });

class UniqueNOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    OpInputList in;
    OP_REQUIRES_OK(ctx, ctx->input_list("inputs", &in));
    const int n = in.size();
    std::vector<const int64_t*> src(n);
    std::vector<int64_t*> uniq(n);
    std::vector<int32_t*> index(n), count(n);
    std::vector<int64_t> lens(n);
    for (int i = 0; i < n; ++i) {
      OP_REQUIRES(ctx, TensorShapeUtils::IsVector(in[i].shape()),
                  errors::InvalidArgument("Input must be a vector"));
      Tensor *u, *x, *k;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, in[i].shape(), &u));
      OP_REQUIRES_OK(ctx, ctx->allocate_output(n + i, in[i].shape(), &x));
      OP_REQUIRES_OK(ctx, ctx->allocate_output(2 * n + i, TensorShape({1}), &k));
      src[i] = reinterpret_cast<const int64_t*>(in[i].flat<int64>().data());
      uniq[i] = reinterpret_cast<int64_t*>(u->flat<int64>().data());
      index[i] = x->flat<int32>().data();
      count[i] = k->flat<int32>().data();
      lens[i] = in[i].NumElements();
    }
    const size_t ws_bytes = hbk_unique_workspace_bytes(n, lens.data());
    Tensor ws;
    OP_REQUIRES_OK(ctx, AllocScratch(ctx, ws_bytes, &ws));
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_unique_n(n, src.data(), lens.data(), uniq.data(), index.data(),
                                               count.data(), ws.flat<int8>().data(), ws_bytes + 16,
                                               StreamOf(ctx))));
  }
};
REGISTER_KERNEL_BUILDER(Name("HbUniqueN").Device(DEVICE_GPU), UniqueNOp);

REGISTER_OP("HbCastN")
    .Output("outputs: N * DstT").Input("inputs: N * SrcT")
    .Attr("N: int >= 1 = 1").Attr("SrcT: {half, float}").Attr("DstT: {half, float}")
    .SetShapeFn([](InferenceContext* c) {
      int64 n;
      TF_RETURN_IF_ERROR(c->GetAttr("N", &n));
      for (int64 i = 0; i < n; ++i) c->set_output(i, c->input(i));
      return Status::OK();
    });

template <typename Src, typename Dst>
class CastNOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    OpInputList in;
    OP_REQUIRES_OK(ctx, ctx->input_list("inputs", &in));
    const int n = in.size();
    std::vector<const void*> src(n);
    std::vector<void*> dst(n);
    std::vector<int64_t> lens(n);
    for (int i = 0; i < n; ++i) {
      Tensor* o;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, in[i].shape(), &o));
      src[i] = in[i].flat<Src>().data();
      dst[i] = o->flat<Dst>().data();
      lens[i] = in[i].NumElements();
    }
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_cast_n(n, HbkType<Src>::v, HbkType<Dst>::v, src.data(),
                                             lens.data(), dst.data(), StreamOf(ctx))));
  }
};
REGISTER_KERNEL_BUILDER(Name("HbCastN").Device(DEVICE_GPU).TypeConstraint<float>("SrcT")
                            .TypeConstraint<Eigen::half>("DstT"),
                        CastNOp<float, Eigen::half>);
REGISTER_KERNEL_BUILDER(Name("HbCastN").Device(DEVICE_GPU).TypeConstraint<Eigen::half>("SrcT")
                            .TypeConstraint<float>("DstT"),
                        CastNOp<Eigen::half, float>);

// ============================================================================================
// HbShardedGroupLookup / HbShardedGroupLookupGrad / HbShardedGroupLookupGradApply (new, additive):
// the whole composition of hbtf/embedding/sharding.py:171-205 for N columns as ONE op per
// direction -- hbk_sharded_lookup_fwd / _bwd, the path every sharded number in profiles/ measures.
// Conventions of the exchange ops they replace (nccl_alltoallv.cc:359-387): first input = the
// communicator resource, N-ary lists, run on the communicator's thread pool because the forward
// waits for the size exchange on the host once.  The forward creates (once, by shared_name) an
// HbShardedPlan resource that owns the hbk_sharded_t, and hands its handle to the gradient op as
// an output: the data dependency also orders the backward behind ITS forward.
// ============================================================================================
class HbShardedPlan : public ResourceBase {
 public:
  HbShardedPlan() : plan_(nullptr) {}
  ~HbShardedPlan() override {
    if (plan_ != nullptr) hbk_sharded_destroy(plan_);
  }
  // (re)creates the plan when the shards it was made for have moved or changed shape
  Status Ensure(hbk_comm_t comm, const std::vector<hbk_sharded_column_t>& cols, int32_t wire_dtype) {
    bool same = plan_ != nullptr && cols.size() == cols_.size();
    for (size_t i = 0; same && i < cols.size(); ++i) {
      same = cols[i].shard == cols_[i].shard && cols[i].rows_local == cols_[i].rows_local &&
             cols[i].dim == cols_[i].dim && cols[i].accum == cols_[i].accum;
    }
    if (same) return Status::OK();
    if (plan_ != nullptr) TF_RETURN_IF_ERROR(HbkStatus(hbk_sharded_destroy(plan_)));
    plan_ = nullptr;
    TF_RETURN_IF_ERROR(HbkStatus(
        hbk_sharded_create(&plan_, comm, static_cast<int32_t>(cols.size()), cols.data(), wire_dtype)));
    cols_ = cols;
    return Status::OK();
  }
  hbk_sharded_t plan() const { return plan_; }
  const std::vector<hbk_sharded_column_t>& cols() const { return cols_; }
  string DebugString() const override { return "HbShardedPlan(libhbk_core)"; }

 private:
  hbk_sharded_t plan_;
  std::vector<hbk_sharded_column_t> cols_;
};

REGISTER_OP("HbShardedGroupLookup")
    .Output("outputs: N * float").Output("plan: resource")
    .Input("handle: resource").Input("shards: N * float").Input("accums: M * float")
    .Input("ids: N * int64").Input("row_splits: N * int32")
    .Attr("N: int >= 1").Attr("M: int >= 0 = 0")
    .Attr("buckets: list(int)").Attr("combiners: list(int)").Attr("ragged: list(bool)")
    .Attr("dedup: list(bool) = []").Attr("hot_rows: list(bool) = []")
    .Attr("wire_dtype: " HB_WIRE_DTYPES " = DT_FLOAT")
    .Attr("container: string = ''").Attr("shared_name: string")
    .SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      int64 n, m;
      TF_RETURN_IF_ERROR(c->GetAttr("N", &n));
      TF_RETURN_IF_ERROR(c->GetAttr("M", &m));
      std::vector<bool> ragged;
      TF_RETURN_IF_ERROR(c->GetAttr("ragged", &ragged));
      for (int64 i = 0; i < n; ++i) {   // inputs: handle, shards [1, 1+n), accums, ids, row_splits
        shape_inference::ShapeHandle ids = c->input(1 + n + m + i);
        shape_inference::ShapeHandle splits = c->input(1 + 2 * n + m + i);
        shape_inference::DimensionHandle segs = c->Dim(ids, 0);
        if (i < static_cast<int64>(ragged.size()) && ragged[i]) {
          TF_RETURN_IF_ERROR(c->Subtract(c->Dim(splits, 0), 1, &segs));
        }
        c->set_output(i, c->Matrix(segs, c->Dim(c->input(1 + i), 1)));
      }
      c->set_output(n, c->Scalar());
      return Status::OK();
    });

class ShardedGroupLookupOp : public CollectiveAsyncOp {
 public:
  explicit ShardedGroupLookupOp(OpKernelConstruction* ctx) : CollectiveAsyncOp(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("buckets", &buckets_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("combiners", &combiners_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("ragged", &ragged_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("dedup", &dedup_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("hot_rows", &hot_rows_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("container", &container_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("shared_name", &name_));
    DataType wire;
    OP_REQUIRES_OK(ctx, ctx->GetAttr("wire_dtype", &wire));
    wire_dtype_ = wire == DT_HALF ? HBK_HALF : HBK_FLOAT;
  }
  void Run(OpKernelContext* ctx, HbNcclCollective* coll) override {
    OpInputList shards, accums, ids, splits;
    OP_REQUIRES_OK(ctx, ctx->input_list("shards", &shards));
    OP_REQUIRES_OK(ctx, ctx->input_list("accums", &accums));
    OP_REQUIRES_OK(ctx, ctx->input_list("ids", &ids));
    OP_REQUIRES_OK(ctx, ctx->input_list("row_splits", &splits));
    const int n = shards.size();
    OP_REQUIRES(ctx, static_cast<int>(buckets_.size()) == n &&
                         static_cast<int>(combiners_.size()) == n &&
                         static_cast<int>(ragged_.size()) == n,
                errors::InvalidArgument("buckets, combiners and ragged must have N entries"));
    OP_REQUIRES(ctx, accums.size() == 0 || accums.size() == n,
                errors::InvalidArgument("accums: none, or one accumulator per shard"));
    std::vector<hbk_sharded_column_t> cols(n);
    for (int i = 0; i < n; ++i) {
      std::memset(&cols[i], 0, sizeof(cols[i]));
      cols[i].shard = shards[i].flat<float>().data();
      cols[i].rows_local = shards[i].dim_size(0);
      cols[i].dim = static_cast<int32_t>(shards[i].dim_size(1));
      cols[i].combiner = combiners_[i];
      cols[i].bucket = buckets_[i];
      cols[i].accum = accums.size() ? const_cast<float*>(accums[i].flat<float>().data()) : nullptr;
      cols[i].hot_rows = i < static_cast<int>(hot_rows_.size()) && hot_rows_[i] ? 1 : 0;
      cols[i].dedup = i < static_cast<int>(dedup_.size()) && dedup_[i] ? 1 : 0;
    }
    const ResourceHandle handle = MakeResourceHandle<HbShardedPlan>(ctx, container_, name_);
    HbShardedPlan* plan = nullptr;
    OP_REQUIRES_OK(ctx, LookupOrCreateResource<HbShardedPlan>(
                            ctx, handle, &plan, [](HbShardedPlan** p) {
                              *p = new HbShardedPlan();
                              return Status::OK();
                            }));
    core::ScopedUnref unref(plan);
    OP_REQUIRES_OK(ctx, plan->Ensure(coll->comm(), cols, wire_dtype_));
    std::vector<const int64_t*> id_ptrs(n);
    std::vector<const int32_t*> split_ptrs(n);
    std::vector<int64_t> n_ids(n), n_seg(n);
    std::vector<float*> outs(n);
    for (int i = 0; i < n; ++i) {
      id_ptrs[i] = reinterpret_cast<const int64_t*>(ids[i].flat<int64>().data());
      n_ids[i] = ids[i].NumElements();
      split_ptrs[i] = ragged_[i] ? splits[i].flat<int32>().data() : nullptr;
      n_seg[i] = ragged_[i] ? splits[i].NumElements() - 1 : n_ids[i];
      Tensor* o;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(i, TensorShape({n_seg[i], shards[i].dim_size(1)}), &o));
      outs[i] = o->flat<float>().data();
    }
    Tensor* plan_out;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(n, TensorShape({}), &plan_out));
    plan_out->scalar<ResourceHandle>()() = handle;
    OP_REQUIRES_OK(ctx, HbkStatus(hbk_sharded_lookup_fwd(plan->plan(), id_ptrs.data(), n_ids.data(),
                                                         split_ptrs.data(), n_seg.data(), outs.data(),
                                                         nullptr, StreamOf(ctx))));
  }

 private:
  std::vector<int64> buckets_;
  std::vector<int32> combiners_;
  std::vector<bool> ragged_, dedup_, hot_rows_;
  string container_, name_;
  int32_t wire_dtype_;
};
REGISTER_KERNEL_BUILDER(Name("HbShardedGroupLookup").Device(DEVICE_GPU).HostMemory("plan"),
                        ShardedGroupLookupOp);

// The backward differentiates the plan's LAST forward (hbk.h): unique_rows / grad_rows have the
// capacity hbk_sharded_owned_ids(plan, c) -- the ids this rank's shard was asked for -- known on
// the host after that forward, so the outputs are sized exactly and no host sync happens here.
REGISTER_OP("HbShardedGroupLookupGrad")
    .Output("unique_rows: N * int64").Output("grad_rows: N * float").Output("n_unique: N * int32")
    .Input("handle: resource").Input("plan: resource").Input("grads: N * float")
    .Attr("N: int >= 1").SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      int64 n;
      TF_RETURN_IF_ERROR(c->GetAttr("N", &n));
      for (int64 i = 0; i < n; ++i) {
        c->set_output(i, c->Vector(InferenceContext::kUnknownDim));
        c->set_output(n + i, c->Matrix(InferenceContext::kUnknownDim, c->Dim(c->input(2 + i), 1)));
        c->set_output(2 * n + i, c->Vector(1));
      }
      return Status::OK();
    });
REGISTER_OP("HbShardedGroupLookupGradApply")
    .Output("n_unique: N * int32")
    .Input("handle: resource").Input("plan: resource").Input("grads: N * float").Input("lr: float")
    .Attr("N: int >= 1").Attr("optimizer: {'sgd', 'adagrad'} = 'sgd'").SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      int64 n;
      TF_RETURN_IF_ERROR(c->GetAttr("N", &n));
      for (int64 i = 0; i < n; ++i) c->set_output(i, c->Vector(1));
      return Status::OK();
    });

template <bool APPLY>
class ShardedGroupLookupGradOp : public CollectiveAsyncOp {
 public:
  explicit ShardedGroupLookupGradOp(OpKernelConstruction* ctx)
      : CollectiveAsyncOp(ctx), apply_(HBK_APPLY_SGD) {
    if (APPLY) {
      string optimizer;
      OP_REQUIRES_OK(ctx, ctx->GetAttr("optimizer", &optimizer));
      OP_REQUIRES_OK(ctx, OptimizerCode(optimizer, &apply_));
    }
  }
  void Run(OpKernelContext* ctx, HbNcclCollective* coll) override {
    HbShardedPlan* plan = nullptr;
    OP_REQUIRES_OK(ctx, LookupResource(ctx, HandleFromInput(ctx, 1), &plan));
    core::ScopedUnref unref(plan);
    OpInputList grads;
    OP_REQUIRES_OK(ctx, ctx->input_list("grads", &grads));
    const int n = grads.size();
    OP_REQUIRES(ctx, plan->plan() != nullptr && static_cast<int>(plan->cols().size()) == n,
                errors::InvalidArgument("plan was made for ", plan->cols().size(), " columns, got ", n,
                                        " gradients"));
    std::vector<const float*> g(n);
    std::vector<int64_t*> urows(n, nullptr);
    std::vector<float*> grows(n, nullptr);
    std::vector<int32_t*> counts(n);
    for (int i = 0; i < n; ++i) {
      OP_REQUIRES(ctx, grads[i].dims() == 2 && grads[i].dim_size(1) == plan->cols()[i].dim,
                  errors::InvalidArgument("gradient ", i, " must be [segments, ", plan->cols()[i].dim, "]"));
      g[i] = grads[i].flat<float>().data();
      Tensor* k;
      if (!APPLY) {
        const int64 cap = hbk_sharded_owned_ids(plan->plan(), i);
        OP_REQUIRES(ctx, cap >= 0, errors::Internal("no forward to differentiate: ", hbk_last_error()));
        Tensor *u, *r;
        OP_REQUIRES_OK(ctx, ctx->allocate_output(i, TensorShape({cap}), &u));
        OP_REQUIRES_OK(ctx, ctx->allocate_output(n + i, TensorShape({cap, plan->cols()[i].dim}), &r));
        urows[i] = reinterpret_cast<int64_t*>(u->flat<int64>().data());
        grows[i] = r->flat<float>().data();
      }
      OP_REQUIRES_OK(ctx, ctx->allocate_output(APPLY ? i : 2 * n + i, TensorShape({1}), &k));
      counts[i] = k->flat<int32>().data();
    }
    float lr = 0.0f;
    if (APPLY) {
      const Tensor* t;
      OP_REQUIRES_OK(ctx, ctx->input("lr", &t));
      lr = t->scalar<float>()();
    }
    if (APPLY) {   // step only: no IndexedSlices are written
      OP_REQUIRES_OK(ctx, HbkStatus(hbk_sharded_lookup_bwd_apply(plan->plan(), g.data(), nullptr, apply_,
                                                               lr, nullptr, nullptr, counts.data(),
                                                               StreamOf(ctx))));
    } else {
      OP_REQUIRES_OK(ctx, HbkStatus(hbk_sharded_lookup_bwd(plan->plan(), g.data(), nullptr, 0.0f,
                                                         urows.data(), grows.data(), counts.data(),
                                                         StreamOf(ctx))));
    }
  }

 private:
  int32_t apply_;
};
REGISTER_KERNEL_BUILDER(
    Name("HbShardedGroupLookupGrad").Device(DEVICE_GPU).HostMemory("plan"),
    ShardedGroupLookupGradOp<false>);
REGISTER_KERNEL_BUILDER(
    Name("HbShardedGroupLookupGradApply").Device(DEVICE_GPU).HostMemory("plan").HostMemory("lr"),
    ShardedGroupLookupGradOp<true>);

}  // namespace hybridbackend
}  // namespace tensorflow
