// Compile-check stub of the slice of TensorFlow's op-kernel API that
// nann_amd/tf_ops/nann_tf_ops.cc (OUR shim) touches.  It exists only so that
// tests/test_tf_shim.py can syntax-check the shim in an image without TensorFlow; it is
// never used to build reference code and implements no behaviour.
#pragma once
#include <cstdint>
#include <functional>
#include <initializer_list>
#include <sstream>
#include <string>
#include <vector>

namespace Eigen { struct half { uint16_t x; }; }

namespace tensorflow {
typedef int32_t int32;
typedef long long int64;
using std::string;

class Status {
 public:
  Status() = default;
  explicit Status(int c, std::string m = "") : code_(c), msg_(std::move(m)) {}
  static Status OK() { return Status(); }
  bool ok() const { return code_ == 0; }
 private:
  int code_ = 0;
  std::string msg_;
};

namespace errors {
template <typename... A> Status make(int code, A&&... a) {
  std::ostringstream os; using sink = int[]; (void)sink{0, ((os << a), 0)...}; return Status(code, os.str());
}
template <typename... A> Status InvalidArgument(A&&... a) { return make(3, a...); }
template <typename... A> Status NotFound(A&&... a) { return make(5, a...); }
template <typename... A> Status Unimplemented(A&&... a) { return make(12, a...); }
template <typename... A> Status Internal(A&&... a) { return make(13, a...); }
}  // namespace errors

class TensorShape {
 public:
  TensorShape() = default;
  TensorShape(std::initializer_list<int64> d) : dims_(d) {}
  int64 dim_size(int i) const { return dims_[i]; }
  int dims() const { return (int)dims_.size(); }
  std::string DebugString() const { return "[...]"; }
 private:
  std::vector<int64> dims_;
};

enum DataType { DT_INVALID = 0, DT_FLOAT = 1, DT_DOUBLE = 2, DT_INT32 = 3, DT_INT64 = 9, DT_HALF = 19 };

template <typename T> struct FlatView { T* p; int64 n; T* data() const { return p; } };
struct StringPieceStub { const char* p; const char* data() const { return p; } };

class Tensor {
 public:
  Tensor() = default;
  Tensor(DataType dt, const TensorShape& s) : dt_(dt), shape_(s) {}
  int64 NumElements() const { return n_; }
  int64 dim_size(int) const { return n_; }
  int dims() const { return shape_.dims(); }
  const TensorShape& shape() const { return shape_; }
  DataType dtype() const { return dt_; }
  StringPieceStub tensor_data() const { return StringPieceStub{static_cast<const char*>(buf_)}; }
  template <typename T> FlatView<T> flat() const { return FlatView<T>{static_cast<T*>(buf_), n_}; }
 private:
  DataType dt_ = DT_INVALID;
  TensorShape shape_;
  void* buf_ = nullptr;
  int64 n_ = 0;
};

class OpKernelConstruction {
 public:
  template <typename T> Status GetAttr(const char*, T*) const { return Status::OK(); }
  void SetStatus(const Status&) {}
};

class OpInputList {
 public:
  const Tensor& operator[](int) const { return t_; }
 private:
  Tensor t_;
};
class OpOutputList {
 public:
  Status allocate(int, const TensorShape&, Tensor** out) { *out = &t_; return Status::OK(); }
 private:
  Tensor t_;
};

class OpKernelContext {
 public:
  const Tensor& input(int) { return t_; }
  Tensor mutable_input(int, bool) { return t_; }
  Status input_list(const char*, OpInputList*) { return Status::OK(); }
  Status output_list(const char*, OpOutputList*) { return Status::OK(); }
  Status allocate_output(int, const TensorShape&, Tensor** out) { *out = &t_; return Status::OK(); }
  void set_output(int, const Tensor&) {}
  void forward_ref_input_to_ref_output(int, int) {}
  void SetStatus(const Status&) {}
 private:
  Tensor t_;
};

class OpKernel {
 public:
  explicit OpKernel(OpKernelConstruction*) {}
  virtual ~OpKernel() = default;
  virtual void Compute(OpKernelContext*) = 0;
  virtual bool IsExpensive() { return true; }
};

class AsyncOpKernel : public OpKernel {
 public:
  using OpKernel::OpKernel;
  typedef std::function<void()> DoneCallback;
  virtual void ComputeAsync(OpKernelContext*, DoneCallback done) = 0;
  void Compute(OpKernelContext*) override {}
};

namespace shape_inference {
struct ShapeHandle {};
struct DimensionHandle {};
class InferenceContext {
 public:
  ShapeHandle input(int) { return {}; }
  Status WithRank(ShapeHandle, int, ShapeHandle*) { return Status::OK(); }
  void set_output(int, ShapeHandle) {}
  ShapeHandle MakeShape(std::initializer_list<DimensionHandle>) { return {}; }
  DimensionHandle UnknownDim() { return {}; }
  template <typename T> Status GetAttr(const char*, T*) const { return Status::OK(); }
  Status MakeShapeFromTensorShape(const TensorShape&, ShapeHandle*) { return Status::OK(); }
};
inline Status UnknownShape(InferenceContext*) { return Status::OK(); }
}  // namespace shape_inference

struct OpDefBuilderStub {
  OpDefBuilderStub& Input(const char*) { return *this; }
  OpDefBuilderStub& Output(const char*) { return *this; }
  OpDefBuilderStub& Attr(const char*) { return *this; }
  OpDefBuilderStub& SetShapeFn(std::function<Status(shape_inference::InferenceContext*)>) { return *this; }
};
struct KernelDefBuilderStub {
  KernelDefBuilderStub& Device(const char*) { return *this; }
  template <typename T> KernelDefBuilderStub& TypeConstraint(const char*) { return *this; }
};
inline KernelDefBuilderStub Name(const char*) { return {}; }
static const char* const DEVICE_CPU = "CPU";

#define TF_STUB_CAT2(a, b) a##b
#define TF_STUB_CAT(a, b) TF_STUB_CAT2(a, b)
#define REGISTER_OP(name) static ::tensorflow::OpDefBuilderStub TF_STUB_CAT(op_stub_, __COUNTER__) = ::tensorflow::OpDefBuilderStub()
#define REGISTER_KERNEL_BUILDER(builder, cls) \
  static ::tensorflow::OpKernel* TF_STUB_CAT(mk_, __COUNTER__)(::tensorflow::OpKernelConstruction* c) { (void)(builder); return new cls(c); }
#define OP_REQUIRES(ctx, cond, status) do { if (!(cond)) { (ctx)->SetStatus(status); return; } } while (0)
#define OP_REQUIRES_OK(ctx, expr) do { ::tensorflow::Status _s = (expr); if (!_s.ok()) { (ctx)->SetStatus(_s); return; } } while (0)
#define OP_REQUIRES_ASYNC(ctx, cond, status, done) do { if (!(cond)) { (ctx)->SetStatus(status); (done)(); return; } } while (0)
#define OP_REQUIRES_OK_ASYNC(ctx, expr, done) do { ::tensorflow::Status _s = (expr); if (!_s.ok()) { (ctx)->SetStatus(_s); (done)(); return; } } while (0)
#define TF_RETURN_IF_ERROR(expr) do { ::tensorflow::Status _s = (expr); if (!_s.ok()) return _s; } while (0)
}  // namespace tensorflow
