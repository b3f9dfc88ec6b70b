// tf_decl_stub.h -- NOT TensorFlow.  Declarations of the TensorFlow 1.15 symbols that
// hb_ops_shim.cc uses, with just enough shape for `hipcc -fsyntax-only` to parse and type-check
// the shim in an image that has no TensorFlow (`make -C integration/tf_shim check`).  It pins
// NOTHING about TensorFlow's behaviour and is never linked or shipped: a real build of the shim
// takes TensorFlow's own headers.  The kernel-registration macro instantiates
// every kernel class, so the bodies of their methods are checked too.
#ifndef HBK_TF_DECL_STUB_H_
#define HBK_TF_DECL_STUB_H_

#include <cstddef>
#include <cstdint>
#include <functional>
#include <initializer_list>
#include <string>
#include <vector>

namespace Eigen {
struct half { uint16_t x; };
struct GpuDevice {
  void* stream() const;
};
}  // namespace Eigen

namespace stream_executor {
struct DeviceMemoryBase {
  DeviceMemoryBase(void* p, size_t n);
};
}  // namespace stream_executor

namespace tensorflow {
namespace se = ::stream_executor;
typedef signed char int8;
typedef unsigned char uint8;
typedef int int32;
typedef unsigned int uint32;
typedef long long int64;
typedef unsigned long long uint64;
using std::string;

namespace error {
enum Code { OK = 0, INVALID_ARGUMENT = 3, UNIMPLEMENTED = 12, INTERNAL = 13 };
}
class Status {
 public:
  Status();
  Status(error::Code code, const string& msg);
  static Status OK();
  bool ok() const;
};
namespace errors {
template <typename... A> Status InvalidArgument(A...);
template <typename... A> Status Internal(A...);
}  // namespace errors

enum DataType { DT_INT8, DT_INT32, DT_INT64, DT_FLOAT, DT_HALF, DT_RESOURCE };

class TensorShape {
 public:
  TensorShape();
  TensorShape(std::initializer_list<int64> dims);
  int dims() const;
  int64 dim_size(int d) const;
  void set_dim(int d, int64 size);
};
class PartialTensorShape {
 public:
  PartialTensorShape();
  PartialTensorShape(std::initializer_list<int64> dims);
  PartialTensorShape Concatenate(const PartialTensorShape& other) const;
  bool AsTensorShape(TensorShape* out) const;
  int dims() const;
  int64 dim_size(int d) const;
};
struct TensorShapeUtils {
  static bool IsVector(const TensorShape& s);
};
struct StringPiece {
  const char* data() const;
};
template <typename T> struct FlatView {
  T* data() const;
};
template <typename T> struct ScalarView {
  T& operator()() const;
};
class Tensor {
 public:
  Tensor();
  const TensorShape& shape() const;
  int64 NumElements() const;
  int dims() const;
  int64 dim_size(int d) const;
  template <typename T> FlatView<T> flat() const;
  template <typename T> ScalarView<T> scalar() const;
  StringPiece tensor_data() const;
};
class OpInputList {
 public:
  int size() const;
  const Tensor& operator[](int i) const;
};
class OpMutableInputList {   // Ref-typed list inputs (op_kernel.h)
 public:
  int size() const;
  Tensor at(int i, bool lock_held);
};

class Stream {
 public:
  Stream& ThenMemcpy(void* host_dst, const se::DeviceMemoryBase& src, uint64 size);
  Stream& ThenMemcpy(se::DeviceMemoryBase* dst, const void* host_src, uint64 size);
  Stream& ThenMemcpy(se::DeviceMemoryBase* dst, const se::DeviceMemoryBase& src, uint64 size);
  Status BlockHostUntilDone();
};
class DeviceContext {
 public:
  Stream* stream() const;
};

class OpKernelConstruction {
 public:
  template <typename T> Status GetAttr(const char* name, T* value) const;
  bool HasAttr(const char* name) const;
  void CtxFailure(const Status& s);
  void CtxFailureWithWarning(const Status& s);
};
class OpKernelContext {
 public:
  const Tensor& input(int i);
  int num_inputs() const;
  Status input_list(const char* name, OpInputList* list);
  Status input(const char* name, const Tensor** tensor);
  Status mutable_input_list(const char* name, OpMutableInputList* list);
  Status allocate_output(int i, const TensorShape& shape, Tensor** out);
  Status allocate_temp(DataType dt, const TensorShape& shape, Tensor* out);
  template <typename D> const D& eigen_device() const;
  DeviceContext* op_device_context();
  void CtxFailure(const Status& s);
  void CtxFailureWithWarning(const Status& s);
};
class OpKernel {
 public:
  explicit OpKernel(OpKernelConstruction* ctx);
  virtual ~OpKernel();
  virtual void Compute(OpKernelContext* ctx) = 0;
};
class AsyncOpKernel : public OpKernel {
 public:
  typedef std::function<void()> DoneCallback;
  explicit AsyncOpKernel(OpKernelConstruction* ctx);
  virtual void ComputeAsync(OpKernelContext* ctx, DoneCallback done) = 0;
  void Compute(OpKernelContext* ctx) final;
};

class ResourceBase {
 public:
  virtual ~ResourceBase();
  virtual string DebugString() const = 0;
  void Unref() const;
};
struct ResourceHandle {};
namespace core {
class ScopedUnref {
 public:
  explicit ScopedUnref(const ResourceBase* o);
  ~ScopedUnref();
};
}  // namespace core
template <typename T> class ResourceHandleOp : public OpKernel {
 public:
  explicit ResourceHandleOp(OpKernelConstruction* ctx);
  void Compute(OpKernelContext* ctx) override;
};
const ResourceHandle& HandleFromInput(OpKernelContext* ctx, int input);
template <typename T> Status LookupResource(OpKernelContext* ctx, const ResourceHandle& h, T** out);
template <typename T> Status CreateResource(OpKernelContext* ctx, const ResourceHandle& h, T* value);
template <typename T>
ResourceHandle MakeResourceHandle(OpKernelContext* ctx, const string& container, const string& name);
template <typename T>
Status LookupOrCreateResource(OpKernelContext* ctx, const ResourceHandle& h, T** value,
                              std::function<Status(T**)> creator);

class Env {
 public:
  static Env* Default();
};
namespace thread {
class ThreadPool {
 public:
  ThreadPool(Env* env, const string& name, int threads);
  ~ThreadPool();
  void Schedule(std::function<void()> fn);
};
}  // namespace thread

namespace shape_inference {
struct ShapeHandle {};
struct DimensionHandle {};
struct DimensionOrConstant {
  DimensionOrConstant(int64 v);
  DimensionOrConstant(DimensionHandle d);
};
class InferenceContext {
 public:
  static constexpr int64 kUnknownDim = -1;
  template <typename T> Status GetAttr(const char* name, T* value) const;
  ShapeHandle input(int i);
  void set_output(int i, ShapeHandle s);
  ShapeHandle Vector(DimensionOrConstant dim);
  ShapeHandle Scalar();
  ShapeHandle Matrix(DimensionOrConstant rows, DimensionOrConstant cols);
  DimensionHandle Dim(ShapeHandle s, int64 idx);
  Status Subtract(DimensionHandle first, DimensionOrConstant second, DimensionHandle* out);
  Status MakeShapeFromPartialTensorShape(const PartialTensorShape& p, ShapeHandle* out);
  Status Concatenate(ShapeHandle a, ShapeHandle b, ShapeHandle* out);
};
Status NoOutputs(InferenceContext* c);
Status ScalarShape(InferenceContext* c);
Status UnchangedShape(InferenceContext* c);
}  // namespace shape_inference

class OpDefBuilderWrapper {
 public:
  explicit OpDefBuilderWrapper(const char* name);
  OpDefBuilderWrapper& Input(const char* spec);
  OpDefBuilderWrapper& Output(const char* spec);
  OpDefBuilderWrapper& Attr(const char* spec);
  OpDefBuilderWrapper& SetIsStateful();
  OpDefBuilderWrapper& SetShapeFn(Status (*fn)(shape_inference::InferenceContext*));
  OpDefBuilderWrapper& Doc(const char* text);
};
class KernelDefBuilder {
 public:
  explicit KernelDefBuilder(const char* op);
  KernelDefBuilder& Device(const char* device);
  template <typename T> KernelDefBuilder& TypeConstraint(const char* attr);
  KernelDefBuilder& HostMemory(const char* arg);
};
inline KernelDefBuilder Name(const char* op) { return KernelDefBuilder(op); }
extern const char* const DEVICE_CPU;
extern const char* const DEVICE_GPU;

// instantiates K: constructor, vtable, hence every virtual method body
template <typename K> int RegisterKernelForCheck(const KernelDefBuilder&) {
  OpKernel* (*make)(OpKernelConstruction*) = [](OpKernelConstruction* c) -> OpKernel* {
    return new K(c);
  };
  return make != nullptr;
}
}  // namespace tensorflow

#define HBK_STUB_CONCAT_(a, b) a##b
#define HBK_STUB_CONCAT(a, b) HBK_STUB_CONCAT_(a, b)
#define REGISTER_OP(name)                                                          \
  static ::tensorflow::OpDefBuilderWrapper& HBK_STUB_CONCAT(hbk_stub_op_, __COUNTER__) = \
      ::tensorflow::OpDefBuilderWrapper(name)
#define REGISTER_KERNEL_BUILDER(builder, ...)                                      \
  static int HBK_STUB_CONCAT(hbk_stub_kernel_, __COUNTER__) =                      \
      ::tensorflow::RegisterKernelForCheck<__VA_ARGS__>(builder)
#define REGISTER_RESOURCE_HANDLE_OP(Type) REGISTER_OP(#Type "HandleOp").Output("resource: resource")
#define TF_RETURN_IF_ERROR(expr)                    \
  do {                                              \
    const ::tensorflow::Status s__ = (expr);        \
    if (!s__.ok()) return s__;                      \
  } while (0)
#define OP_REQUIRES(CTX, EXP, STATUS) \
  do {                                \
    if (!(EXP)) {                     \
      (CTX)->CtxFailure((STATUS));    \
      return;                         \
    }                                 \
  } while (0)
#define OP_REQUIRES_OK(CTX, ...)                       \
  do {                                                 \
    const ::tensorflow::Status s__ = (__VA_ARGS__);    \
    if (!s__.ok()) {                                   \
      (CTX)->CtxFailureWithWarning(s__);               \
      return;                                          \
    }                                                  \
  } while (0)
#define OP_REQUIRES_OK_ASYNC(CTX, STATUS, CALLBACK)    \
  do {                                                 \
    const ::tensorflow::Status s__ = (STATUS);         \
    if (!s__.ok()) {                                   \
      (CTX)->CtxFailureWithWarning(s__);               \
      (CALLBACK)();                                    \
      return;                                          \
    }                                                  \
  } while (0)

#endif  // HBK_TF_DECL_STUB_H_
