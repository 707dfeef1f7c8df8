// A FUNCTIONAL model of the slice of TensorFlow 1.15's op-kernel API that OUR op shim
// (nann_amd/tf_ops/nann_tf_ops.cc) is written against.  TensorFlow is not in this image (no bazel,
// no pip wheel, no network), so "the shim drops into the serving graph" would otherwise rest on code
// that never ran.  With this header the shim is compiled into a real shared object and every
// registered kernel is instantiated from its REGISTER_OP / REGISTER_KERNEL_BUILDER records and run
// on the GPU by tests/test_tf_shim_gpu.py (driver: tests/tf_mock/tfm_driver.cc).
//
// What is modelled, with the behaviour TensorFlow has (checked against the fork's sources, cited):
//   * Tensor: a ref-counted buffer shared by copies (core/framework/tensor.h) -- what makes
//     mutable_input() alias the caller's Ref bitmap and HugeConst's set_output zero-copy;
//     freshly allocated outputs are POISONED (0xCD), not zeroed, as TF's allocators do not zero.
//   * REGISTER_OP: the spec strings are parsed into an OpDef (core/framework/op_def_builder.cc):
//     typed / Ref / list inputs and outputs, attrs with allowed sets, minima and defaults.
//   * REGISTER_KERNEL_BUILDER: (op, device, type constraints) -> factory; kernel lookup fails when no
//     registration matches the node's attrs (core/framework/op_kernel.cc FindKernelRegistration).
//   * OpKernelConstruction::GetAttr type-checks; OpKernelContext checks output dtypes, ref-ness and
//     index ranges; OpInputList / OpOutputList resolve names through the OpDef.
//   * shape_inference::InferenceContext with unknown dims / ranks, enough to RUN every SetShapeFn.
// It is test infrastructure for OUR code only: it is never used to build reference sources.
#pragma once
#include <atomic>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

namespace Eigen {
struct half { uint16_t x; };
}  // namespace Eigen

namespace tensorflow {
typedef int8_t int8;
typedef int32_t int32;
typedef long long int64;
typedef uint64_t uint64;
typedef uint8_t uint8;
using std::string;

[[noreturn]] inline void TfmCheckFail(const char* what, const std::string& detail) {
  std::fprintf(stderr, "tf_mock CHECK failed: %s %s\n", what, detail.c_str());
  std::abort();  // TensorFlow's CHECK aborts the process too
}
#define TFM_CHECK(cond, detail) do { if (!(cond)) ::tensorflow::TfmCheckFail(#cond, (detail)); } while (0)

// ---- Status / errors (core/lib/core/status.h, errors.h, error_codes.proto) --------------------
namespace error {
enum Code { OK = 0, CANCELLED = 1, UNKNOWN = 2, INVALID_ARGUMENT = 3, DEADLINE_EXCEEDED = 4, NOT_FOUND = 5,
            ALREADY_EXISTS = 6, PERMISSION_DENIED = 7, RESOURCE_EXHAUSTED = 8, FAILED_PRECONDITION = 9, ABORTED = 10,
            OUT_OF_RANGE = 11, UNIMPLEMENTED = 12, INTERNAL = 13, UNAVAILABLE = 14, DATA_LOSS = 15 };
}  // namespace error

class Status {
 public:
  Status() = default;
  Status(error::Code c, std::string m) : code_(c), msg_(std::move(m)) {}
  static Status OK() { return Status(); }
  bool ok() const { return code_ == error::OK; }
  error::Code code() const { return code_; }
  const std::string& error_message() const { return msg_; }
  std::string ToString() const { return ok() ? "OK" : ("code " + std::to_string((int)code_) + ": " + msg_); }
  // first error wins (Status::Update)
  void Update(const Status& s) { if (ok()) *this = s; }

 private:
  error::Code code_ = error::OK;
  std::string msg_;
};

namespace errors {
inline void Append(std::ostringstream&) {}
template <typename A, typename... R> void Append(std::ostringstream& os, const A& a, const R&... r) { os << a; Append(os, r...); }
template <typename... A> Status Make(error::Code c, const A&... a) { std::ostringstream os; Append(os, a...); return Status(c, os.str()); }
template <typename... A> Status InvalidArgument(const A&... a) { return Make(error::INVALID_ARGUMENT, a...); }
template <typename... A> Status NotFound(const A&... a) { return Make(error::NOT_FOUND, a...); }
template <typename... A> Status Unimplemented(const A&... a) { return Make(error::UNIMPLEMENTED, a...); }
template <typename... A> Status Internal(const A&... a) { return Make(error::INTERNAL, a...); }
template <typename... A> Status DeadlineExceeded(const A&... a) { return Make(error::DEADLINE_EXCEEDED, a...); }
template <typename... A> Status ResourceExhausted(const A&... a) { return Make(error::RESOURCE_EXHAUSTED, a...); }
template <typename... A> Status FailedPrecondition(const A&... a) { return Make(error::FAILED_PRECONDITION, a...); }
template <typename... A> Status OutOfRange(const A&... a) { return Make(error::OUT_OF_RANGE, a...); }
template <typename... A> Status Unavailable(const A&... a) { return Make(error::UNAVAILABLE, a...); }
}  // namespace errors

// ---- DataType (core/framework/types.proto) ------------------------------------------------------
enum DataType { DT_INVALID = 0, DT_FLOAT = 1, DT_DOUBLE = 2, DT_INT32 = 3, DT_UINT8 = 4, DT_INT16 = 5, DT_INT8 = 6,
                DT_STRING = 7, DT_INT64 = 9, DT_BOOL = 10, DT_HALF = 19,
                DT_FLOAT_REF = 101, DT_INT32_REF = 103, DT_INT64_REF = 109, DT_HALF_REF = 119 };
typedef std::vector<DataType> DataTypeVector;
inline bool IsRefType(DataType dt) { return (int)dt > 100; }
inline DataType MakeRefType(DataType dt) { return IsRefType(dt) ? dt : (DataType)((int)dt + 100); }
inline DataType RemoveRefType(DataType dt) { return IsRefType(dt) ? (DataType)((int)dt - 100) : dt; }
inline int DataTypeSize(DataType dt) {
  switch (RemoveRefType(dt)) {
    case DT_FLOAT: case DT_INT32: return 4;
    case DT_DOUBLE: case DT_INT64: return 8;
    case DT_HALF: case DT_INT16: return 2;
    case DT_UINT8: case DT_INT8: case DT_BOOL: return 1;
    default: return 0;
  }
}
inline std::string DataTypeString(DataType dt) {
  const std::string ref = IsRefType(dt) ? "_ref" : "";
  switch (RemoveRefType(dt)) {
    case DT_FLOAT: return "float" + ref;
    case DT_DOUBLE: return "double" + ref;
    case DT_INT32: return "int32" + ref;
    case DT_UINT8: return "uint8" + ref;
    case DT_INT16: return "int16" + ref;
    case DT_INT8: return "int8" + ref;
    case DT_STRING: return "string" + ref;
    case DT_INT64: return "int64" + ref;
    case DT_BOOL: return "bool" + ref;
    case DT_HALF: return "half" + ref;
    default: return "invalid";
  }
}
// names accepted in op specs (op_def_builder.cc: "float" == "float32", "half" == "float16" ...)
inline bool DataTypeFromName(const std::string& s, DataType* dt) {
  static const std::pair<const char*, DataType> k[] = {
      {"float", DT_FLOAT}, {"float32", DT_FLOAT}, {"double", DT_DOUBLE}, {"float64", DT_DOUBLE}, {"int32", DT_INT32},
      {"uint8", DT_UINT8}, {"int16", DT_INT16}, {"int8", DT_INT8}, {"string", DT_STRING}, {"int64", DT_INT64},
      {"bool", DT_BOOL}, {"half", DT_HALF}, {"float16", DT_HALF}};
  for (const auto& e : k)
    if (s == e.first) { *dt = e.second; return true; }
  return false;
}
template <typename T> struct DataTypeToEnum;
template <> struct DataTypeToEnum<float> { enum : int { value = DT_FLOAT }; static constexpr DataType v() { return DT_FLOAT; } };
template <> struct DataTypeToEnum<double> { enum : int { value = DT_DOUBLE }; static constexpr DataType v() { return DT_DOUBLE; } };
template <> struct DataTypeToEnum<int32> { enum : int { value = DT_INT32 }; static constexpr DataType v() { return DT_INT32; } };
template <> struct DataTypeToEnum<int64> { enum : int { value = DT_INT64 }; static constexpr DataType v() { return DT_INT64; } };
template <> struct DataTypeToEnum<int8> { enum : int { value = DT_INT8 }; static constexpr DataType v() { return DT_INT8; } };
template <> struct DataTypeToEnum<uint8> { enum : int { value = DT_UINT8 }; static constexpr DataType v() { return DT_UINT8; } };
template <> struct DataTypeToEnum<bool> { enum : int { value = DT_BOOL }; static constexpr DataType v() { return DT_BOOL; } };
template <> struct DataTypeToEnum<Eigen::half> { enum : int { value = DT_HALF }; static constexpr DataType v() { return DT_HALF; } };

// ---- TensorShape ---------------------------------------------------------------------------------
class TensorShape {
 public:
  TensorShape() = default;
  TensorShape(std::initializer_list<int64> d) : dims_(d) { Check(); }
  explicit TensorShape(const std::vector<int64>& d) : dims_(d) { Check(); }
  int dims() const { return (int)dims_.size(); }
  int64 dim_size(int i) const {
    TFM_CHECK(i >= 0 && i < dims(), "TensorShape::dim_size(" + std::to_string(i) + ") of rank " + std::to_string(dims()));
    return dims_[(size_t)i];
  }
  void AddDim(int64 d) { dims_.push_back(d); Check(); }
  int64 num_elements() const { int64 n = 1; for (int64 d : dims_) n *= d; return n; }
  bool operator==(const TensorShape& o) const { return dims_ == o.dims_; }
  bool operator!=(const TensorShape& o) const { return dims_ != o.dims_; }
  std::string DebugString() const {
    std::string s = "[";
    for (size_t i = 0; i < dims_.size(); ++i) s += (i ? "," : "") + std::to_string(dims_[i]);
    return s + "]";
  }
  const std::vector<int64>& dim_sizes() const { return dims_; }

 private:
  void Check() const { for (int64 d : dims_) TFM_CHECK(d >= 0, "negative dimension in a TensorShape: " + DebugString()); }
  std::vector<int64> dims_;
};

// ---- Tensor: shared, ref-counted buffer ---------------------------------------------------------
struct TensorBuffer {
  void* data = nullptr;
  size_t bytes = 0;
  bool owned = false;
  ~TensorBuffer() { if (owned) std::free(data); }
};

class StringPiece {
 public:
  StringPiece(const char* p, size_t n) : p_(p), n_(n) {}
  const char* data() const { return p_; }
  size_t size() const { return n_; }

 private:
  const char* p_;
  size_t n_;
};

template <typename T> class TensorView {  // what flat<T>() / vec<T>() / scalar<T>() return (Eigen::TensorMap in TF)
 public:
  TensorView(T* p, int64 n) : p_(p), n_(n) {}
  T* data() const { return p_; }
  int64 size() const { return n_; }
  int64 dimension(int) const { return n_; }
  T& operator()(int64 i) const { return p_[i]; }
  T& operator()() const { return p_[0]; }

 private:
  T* p_;
  int64 n_;
};
template <typename T> struct TTypes {
  typedef TensorView<T> Flat;
  typedef TensorView<const T> ConstFlat;
  typedef TensorView<T> Vec;
  typedef TensorView<const T> ConstVec;
};

class Tensor {
 public:
  Tensor() = default;
  // allocates; contents are POISON, not zeros (TF's allocators do not clear memory)
  Tensor(DataType dt, const TensorShape& s) : dt_(dt), shape_(s) {
    TFM_CHECK(DataTypeSize(dt) > 0, "Tensor of unsupported dtype " + DataTypeString(dt));
    const size_t bytes = (size_t)s.num_elements() * (size_t)DataTypeSize(dt);
    buf_ = std::make_shared<TensorBuffer>();
    buf_->bytes = bytes;
    buf_->owned = true;
    if (bytes) {
      if (posix_memalign(&buf_->data, 64, (bytes + 63) / 64 * 64) != 0) TfmCheckFail("posix_memalign", "out of memory");
      std::memset(buf_->data, 0xCD, bytes);
    }
  }
  // wraps memory the caller keeps alive (a numpy array of the test): no copy, so in-place updates are visible to it
  static Tensor Borrow(DataType dt, const TensorShape& s, void* data) {
    Tensor t;
    t.dt_ = dt;
    t.shape_ = s;
    t.buf_ = std::make_shared<TensorBuffer>();
    t.buf_->data = data;
    t.buf_->bytes = (size_t)s.num_elements() * (size_t)DataTypeSize(dt);
    return t;
  }
  DataType dtype() const { return dt_; }
  const TensorShape& shape() const { return shape_; }
  int dims() const { return shape_.dims(); }
  int64 dim_size(int i) const { return shape_.dim_size(i); }
  int64 NumElements() const { return shape_.num_elements(); }
  size_t TotalBytes() const { return buf_ ? buf_->bytes : 0; }
  bool IsInitialized() const { return buf_ != nullptr; }
  bool SharesBufferWith(const Tensor& o) const { return buf_ && o.buf_ && buf_->data == o.buf_->data; }
  StringPiece tensor_data() const { return StringPiece(static_cast<const char*>(base()), TotalBytes()); }

  template <typename T> TensorView<T> flat() { CheckType<T>(); return TensorView<T>(static_cast<T*>(base()), NumElements()); }
  template <typename T> TensorView<const T> flat() const { CheckType<T>(); return TensorView<const T>(static_cast<const T*>(base()), NumElements()); }
  template <typename T> TensorView<T> vec() { CheckRank(1); return flat<T>(); }
  template <typename T> TensorView<const T> vec() const { CheckRank(1); return flat<T>(); }
  template <typename T> TensorView<T> scalar() { CheckOne(); return flat<T>(); }
  template <typename T> TensorView<const T> scalar() const { CheckOne(); return flat<T>(); }

 private:
  void* base() const { return buf_ ? buf_->data : nullptr; }
  template <typename T> void CheckType() const {
    TFM_CHECK(DataTypeToEnum<typename std::remove_const<T>::type>::v() == RemoveRefType(dt_),
              "flat<T>() on a tensor of dtype " + DataTypeString(dt_));
  }
  void CheckRank(int r) const { TFM_CHECK(dims() == r, "vec<T>() on a tensor of shape " + shape_.DebugString()); }
  void CheckOne() const { TFM_CHECK(NumElements() == 1, "scalar<T>() on a tensor of shape " + shape_.DebugString()); }
  DataType dt_ = DT_INVALID;
  TensorShape shape_;
  std::shared_ptr<TensorBuffer> buf_;
};

// ---- attrs ------------------------------------------------------------------------------------------
struct AttrValue {
  enum Kind { kNone, kBool, kInt, kFloat, kString, kType, kShape, kListString, kListType, kListInt } kind = kNone;
  bool b = false;
  int64 i = 0;
  float f = 0.f;
  std::string s;
  DataType type = DT_INVALID;
  TensorShape shape;
  std::vector<std::string> list_s;
  std::vector<DataType> list_type;
  std::vector<int64> list_i;
};
typedef std::map<std::string, AttrValue> AttrMap;

inline const char* AttrKindName(AttrValue::Kind k) {
  switch (k) {
    case AttrValue::kBool: return "bool";
    case AttrValue::kInt: return "int";
    case AttrValue::kFloat: return "float";
    case AttrValue::kString: return "string";
    case AttrValue::kType: return "type";
    case AttrValue::kShape: return "shape";
    case AttrValue::kListString: return "list(string)";
    case AttrValue::kListType: return "list(type)";
    case AttrValue::kListInt: return "list(int)";
    default: return "none";
  }
}

struct AttrReader {  // shared by OpKernelConstruction and InferenceContext (node_def_util.h GetNodeAttr)
  const AttrMap* attrs = nullptr;
  Status Find(const std::string& name, AttrValue::Kind kind, const AttrValue** out) const {
    auto it = attrs->find(name);
    if (it == attrs->end()) return errors::NotFound("No attr named '", name, "' in NodeDef");
    if (it->second.kind != kind)
      return errors::InvalidArgument("AttrValue had value with type '", AttrKindName(it->second.kind), "' when '",
                                     AttrKindName(kind), "' expected for attr '", name, "'");
    *out = &it->second;
    return Status::OK();
  }
  Status Get(const std::string& n, bool* v) const { const AttrValue* a = nullptr; Status s = Find(n, AttrValue::kBool, &a); if (s.ok()) *v = a->b; return s; }
  Status Get(const std::string& n, int64* v) const { const AttrValue* a = nullptr; Status s = Find(n, AttrValue::kInt, &a); if (s.ok()) *v = a->i; return s; }
  Status Get(const std::string& n, int32* v) const {
    const AttrValue* a = nullptr;
    Status s = Find(n, AttrValue::kInt, &a);
    if (!s.ok()) return s;
    if (a->i < INT32_MIN || a->i > INT32_MAX) return errors::InvalidArgument("Attr ", n, " has value ", a->i, " out of range for an int32");
    *v = (int32)a->i;
    return s;
  }
  Status Get(const std::string& n, float* v) const { const AttrValue* a = nullptr; Status s = Find(n, AttrValue::kFloat, &a); if (s.ok()) *v = a->f; return s; }
  Status Get(const std::string& n, std::string* v) const { const AttrValue* a = nullptr; Status s = Find(n, AttrValue::kString, &a); if (s.ok()) *v = a->s; return s; }
  Status Get(const std::string& n, DataType* v) const { const AttrValue* a = nullptr; Status s = Find(n, AttrValue::kType, &a); if (s.ok()) *v = a->type; return s; }
  Status Get(const std::string& n, TensorShape* v) const { const AttrValue* a = nullptr; Status s = Find(n, AttrValue::kShape, &a); if (s.ok()) *v = a->shape; return s; }
  Status Get(const std::string& n, std::vector<std::string>* v) const { const AttrValue* a = nullptr; Status s = Find(n, AttrValue::kListString, &a); if (s.ok()) *v = a->list_s; return s; }
  Status Get(const std::string& n, std::vector<DataType>* v) const { const AttrValue* a = nullptr; Status s = Find(n, AttrValue::kListType, &a); if (s.ok()) *v = a->list_type; return s; }
  Status Get(const std::string& n, std::vector<int64>* v) const { const AttrValue* a = nullptr; Status s = Find(n, AttrValue::kListInt, &a); if (s.ok()) *v = a->list_i; return s; }
};

// ---- shape inference (core/framework/shape_inference.h) -------------------------------------------
namespace shape_inference {
struct Shape {
  bool known_rank = false;
  std::vector<int64> dims;  // -1 = unknown
};
class DimensionHandle {
 public:
  DimensionHandle() = default;
  DimensionHandle(int64 v) : v_(v) {}  // NOLINT: TF has the same implicit ctor
  int64 v_ = -1;
};
class ShapeHandle {
 public:
  ShapeHandle() = default;
  explicit ShapeHandle(const Shape* s) : s_(s) {}
  const Shape* s_ = nullptr;
};

class InferenceContext {
 public:
  enum : int { kUnknownDim = -1, kUnknownRank = -1 };
  InferenceContext(const AttrMap* attrs, const std::vector<Shape>& inputs, int n_outputs) : outputs_((size_t)n_outputs) {
    reader_.attrs = attrs;
    for (const Shape& s : inputs) inputs_.push_back(Keep(s));
  }
  int num_inputs() const { return (int)inputs_.size(); }
  int num_outputs() const { return (int)outputs_.size(); }
  ShapeHandle input(int i) const { TFM_CHECK(i >= 0 && i < num_inputs(), "InferenceContext::input " + std::to_string(i)); return inputs_[(size_t)i]; }
  ShapeHandle output(int i) const { return outputs_[(size_t)i]; }
  void set_output(int i, ShapeHandle s) { TFM_CHECK(i >= 0 && i < num_outputs(), "InferenceContext::set_output " + std::to_string(i)); outputs_[(size_t)i] = s; }
  bool RankKnown(ShapeHandle s) const { return s.s_ && s.s_->known_rank; }
  int32 Rank(ShapeHandle s) const { return RankKnown(s) ? (int32)s.s_->dims.size() : kUnknownRank; }
  DimensionHandle Dim(ShapeHandle s, int64 idx) const {
    if (!RankKnown(s)) return UnknownDim();
    const int64 r = (int64)s.s_->dims.size();
    if (idx < 0) idx += r;
    TFM_CHECK(idx >= 0 && idx < r, "InferenceContext::Dim index");
    return DimensionHandle(s.s_->dims[(size_t)idx]);
  }
  static int64 Value(DimensionHandle d) { return d.v_; }
  static bool ValueKnown(DimensionHandle d) { return d.v_ >= 0; }
  DimensionHandle UnknownDim() const { return DimensionHandle((int64)-1); }
  DimensionHandle MakeDim(int64 v) const { return DimensionHandle(v); }
  ShapeHandle UnknownShape() { return Keep(Shape()); }
  ShapeHandle Scalar() { return MakeShape({}); }
  ShapeHandle Vector(DimensionHandle d) { return MakeShape({d}); }
  ShapeHandle MakeShape(std::initializer_list<DimensionHandle> dims) {
    Shape s;
    s.known_rank = true;
    for (const DimensionHandle& d : dims) s.dims.push_back(d.v_);
    return Keep(s);
  }
  ShapeHandle MakeShape(const std::vector<DimensionHandle>& dims) {
    Shape s;
    s.known_rank = true;
    for (const DimensionHandle& d : dims) s.dims.push_back(d.v_);
    return Keep(s);
  }
  Status MakeShapeFromTensorShape(const TensorShape& ts, ShapeHandle* out) {
    Shape s;
    s.known_rank = true;
    for (int i = 0; i < ts.dims(); ++i) s.dims.push_back(ts.dim_size(i));
    *out = Keep(s);
    return Status::OK();
  }
  Status WithRank(ShapeHandle s, int64 rank, ShapeHandle* out) {
    if (!RankKnown(s)) {
      Shape r;
      r.known_rank = true;
      r.dims.assign((size_t)rank, (int64)-1);
      *out = Keep(r);
      return Status::OK();
    }
    if ((int64)s.s_->dims.size() != rank)
      return errors::InvalidArgument("Shape must be rank ", rank, " but is rank ", s.s_->dims.size());
    *out = s;
    return Status::OK();
  }
  Status WithRankAtMost(ShapeHandle s, int64 rank, ShapeHandle* out) {
    if (RankKnown(s) && (int64)s.s_->dims.size() > rank)
      return errors::InvalidArgument("Shape must be at most rank ", rank, " but is rank ", s.s_->dims.size());
    *out = s;
    return Status::OK();
  }
  Status WithRankAtLeast(ShapeHandle s, int64 rank, ShapeHandle* out) {
    if (RankKnown(s) && (int64)s.s_->dims.size() < rank)
      return errors::InvalidArgument("Shape must be at least rank ", rank, " but is rank ", s.s_->dims.size());
    *out = s;
    return Status::OK();
  }
  Status Subshape(ShapeHandle s, int64 start, int64 end, ShapeHandle* out) {
    if (!RankKnown(s)) { *out = UnknownShape(); return Status::OK(); }
    const int64 r = (int64)s.s_->dims.size();
    if (start < 0) start += r;
    if (end < 0) end += r;
    if (start < 0 || end > r || start > end) return errors::InvalidArgument("Subshape out of range");
    Shape o;
    o.known_rank = true;
    o.dims.assign(s.s_->dims.begin() + start, s.s_->dims.begin() + end);
    *out = Keep(o);
    return Status::OK();
  }
  Status Concatenate(ShapeHandle a, ShapeHandle b, ShapeHandle* out) {
    if (!RankKnown(a) || !RankKnown(b)) { *out = UnknownShape(); return Status::OK(); }
    Shape o;
    o.known_rank = true;
    o.dims = a.s_->dims;
    o.dims.insert(o.dims.end(), b.s_->dims.begin(), b.s_->dims.end());
    *out = Keep(o);
    return Status::OK();
  }
  Status ReplaceDim(ShapeHandle s, int64 idx, DimensionHandle d, ShapeHandle* out) {
    if (!RankKnown(s)) { *out = UnknownShape(); return Status::OK(); }
    Shape o = *s.s_;
    const int64 r = (int64)o.dims.size();
    if (idx < 0) idx += r;
    if (idx < 0 || idx >= r) return errors::InvalidArgument("Out of range dim_index ", idx, " for shape with ", r, " dimensions");
    o.dims[(size_t)idx] = d.v_;
    *out = Keep(o);
    return Status::OK();
  }
  // value of a scalar input when it is a graph constant; the mock's graphs have none
  Status MakeDimForScalarInput(int, DimensionHandle* out) { *out = UnknownDim(); return Status::OK(); }
  template <typename T> Status GetAttr(const std::string& name, T* v) const { return reader_.Get(name, v); }

 private:
  ShapeHandle Keep(const Shape& s) { arena_.push_back(std::unique_ptr<Shape>(new Shape(s))); return ShapeHandle(arena_.back().get()); }
  AttrReader reader_;
  std::vector<std::unique_ptr<Shape>> arena_;
  std::vector<ShapeHandle> inputs_, outputs_;
};

inline Status UnknownShape(InferenceContext* c) {
  for (int i = 0; i < c->num_outputs(); ++i) c->set_output(i, c->UnknownShape());
  return Status::OK();
}
}  // namespace shape_inference

// ---- OpDef: REGISTER_OP's spec strings, parsed (core/framework/op_def_builder.cc) -----------------
struct ArgDef {
  std::string name, spec;     // spec = the text after "name:", as written
  bool is_ref = false;
  DataType fixed = DT_INVALID;  // a literal dtype ...
  std::string type_attr;        // ... or a `type` attr ...
  std::string type_list_attr;   // ... or a `list(type)` attr (the arg is a LIST of tensors)
};
struct AttrDef {
  std::string name, spec, type;  // type: bool | int | float | string | type | shape | list(string) | list(type) | list(int)
  std::vector<DataType> allowed; // type / list(type) attrs with a {..} set
  bool has_min = false;
  int64 min = 0;
  bool has_default = false;
  AttrValue def;
};
struct OpDef {
  std::string name;
  std::vector<ArgDef> inputs, outputs;
  std::vector<AttrDef> attrs;
  std::function<Status(shape_inference::InferenceContext*)> shape_fn;
  std::string error;  // a spec string that did not parse
  const AttrDef* FindAttr(const std::string& n) const {
    for (const auto& a : attrs) if (a.name == n) return &a;
    return nullptr;
  }
};

namespace spec {
inline std::string Trim(const std::string& s) {
  size_t b = 0, e = s.size();
  while (b < e && std::isspace((unsigned char)s[b])) ++b;
  while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
  return s.substr(b, e - b);
}
inline std::vector<std::string> Split(const std::string& s, char sep) {
  std::vector<std::string> out;
  std::string cur;
  for (char c : s) { if (c == sep) { out.push_back(Trim(cur)); cur.clear(); } else cur.push_back(c); }
  out.push_back(Trim(cur));
  return out;
}
inline bool ParseTypeSet(const std::string& body, std::vector<DataType>* out) {  // "int32, int64"
  for (const std::string& n : Split(body, ',')) {
    DataType dt;
    if (!DataTypeFromName(n, &dt)) return false;
    out->push_back(dt);
  }
  return true;
}
}  // namespace spec

class OpDefBuilder {
 public:
  explicit OpDefBuilder(const char* name) { def_.name = name; }
  OpDefBuilder& Input(const char* s) { ParseArg(s, &def_.inputs); return *this; }
  OpDefBuilder& Output(const char* s) { ParseArg(s, &def_.outputs); return *this; }
  OpDefBuilder& Attr(const char* s) { ParseAttr(s); return *this; }
  OpDefBuilder& SetShapeFn(std::function<Status(shape_inference::InferenceContext*)> fn) { def_.shape_fn = std::move(fn); return *this; }
  OpDefBuilder& Doc(const char*) { return *this; }
  const OpDef& def() const { return def_; }

 private:
  void Fail(const std::string& m) { if (def_.error.empty()) def_.error = m; }
  void ParseArg(const std::string& s, std::vector<ArgDef>* out) {
    const size_t colon = s.find(':');
    if (colon == std::string::npos) return Fail("arg spec without ':': " + s);
    ArgDef a;
    a.name = spec::Trim(s.substr(0, colon));
    a.spec = spec::Trim(s.substr(colon + 1));
    std::string t = a.spec;
    if (t.compare(0, 3, "Ref") == 0) {  // "Ref(int32)" / "Ref (int32)"
      const size_t l = t.find('('), r = t.rfind(')');
      if (l == std::string::npos || r == std::string::npos || r < l) return Fail("bad Ref spec: " + s);
      a.is_ref = true;
      t = spec::Trim(t.substr(l + 1, r - l - 1));
    }
    if (!DataTypeFromName(t, &a.fixed)) a.type_attr = t;  // resolved against the attrs in Finalize
    out->push_back(a);
  }
  void ParseAttr(const std::string& s) {
    const size_t colon = s.find(':');
    if (colon == std::string::npos) return Fail("attr spec without ':': " + s);
    AttrDef a;
    a.name = spec::Trim(s.substr(0, colon));
    a.spec = spec::Trim(s.substr(colon + 1));
    std::string rest = a.spec, def;
    // default: the text after the LAST '=' that is not part of ">="
    for (size_t i = rest.size(); i-- > 0;) {
      if (rest[i] == '=' && (i == 0 || rest[i - 1] != '>')) { def = spec::Trim(rest.substr(i + 1)); rest = spec::Trim(rest.substr(0, i)); a.has_default = true; break; }
    }
    const size_t ge = rest.find(">=");
    if (ge != std::string::npos) {
      a.has_min = true;
      a.min = std::strtoll(rest.c_str() + ge + 2, nullptr, 10);
      rest = spec::Trim(rest.substr(0, ge));
    }
    if (rest.size() >= 2 && rest.front() == '{' && rest.back() == '}') {  // "{int32, int64}": a type attr with a set
      a.type = "type";
      if (!spec::ParseTypeSet(rest.substr(1, rest.size() - 2), &a.allowed)) return Fail("bad type set: " + s);
    } else if (rest.compare(0, 5, "list(") == 0 && rest.back() == ')') {
      std::string inner = spec::Trim(rest.substr(5, rest.size() - 6));
      if (inner.size() >= 2 && inner.front() == '{' && inner.back() == '}') {
        a.type = "list(type)";
        if (!spec::ParseTypeSet(inner.substr(1, inner.size() - 2), &a.allowed)) return Fail("bad type set: " + s);
      } else if (inner == "string" || inner == "type" || inner == "int") a.type = "list(" + inner + ")";
      else return Fail("unsupported list attr: " + s);
    } else if (rest == "bool" || rest == "int" || rest == "float" || rest == "string" || rest == "type" || rest == "shape") {
      a.type = rest;
    } else {
      return Fail("unsupported attr type: " + s);
    }
    if (a.has_default) {
      AttrValue& v = a.def;
      if (a.type == "bool") { v.kind = AttrValue::kBool; if (def != "true" && def != "false") return Fail("bad bool default: " + s); v.b = def == "true"; }
      else if (a.type == "int") { v.kind = AttrValue::kInt; v.i = std::strtoll(def.c_str(), nullptr, 10); }
      else if (a.type == "float") { v.kind = AttrValue::kFloat; v.f = std::strtof(def.c_str(), nullptr); }
      else if (a.type == "string") {
        v.kind = AttrValue::kString;
        if (def.size() < 2 || (def.front() != '\'' && def.front() != '"') || def.back() != def.front()) return Fail("bad string default: " + s);
        v.s = def.substr(1, def.size() - 2);
      } else if (a.type == "type") { v.kind = AttrValue::kType; if (!DataTypeFromName(def, &v.type)) return Fail("bad type default: " + s); }
      else return Fail("default of a " + a.type + " attr is not modelled: " + s);
    }
    def_.attrs.push_back(a);
  }
  OpDef def_;
};

class OpRegistry {
 public:
  static OpRegistry& Global() { static OpRegistry r; return r; }
  void Register(const OpDef& d) {
    OpDef def = d;
    // resolve "T" / "InT" of the args against the attrs
    for (std::vector<ArgDef>* args : {&def.inputs, &def.outputs})
      for (ArgDef& a : *args) {
        if (a.fixed != DT_INVALID) continue;
        const AttrDef* at = def.FindAttr(a.type_attr);
        if (!at) { if (def.error.empty()) def.error = "arg '" + a.name + "' names unknown attr '" + a.type_attr + "'"; continue; }
        if (at->type == "list(type)") { a.type_list_attr = a.type_attr; a.type_attr.clear(); }
        else if (at->type != "type" && def.error.empty()) def.error = "arg '" + a.name + "': attr '" + at->name + "' is not a type";
      }
    std::lock_guard<std::mutex> lk(mu_);
    if (ops_.count(def.name)) duplicate_.push_back(def.name);  // TF: "Op with name X is already registered"
    ops_[def.name] = def;
    order_.push_back(def.name);
  }
  const OpDef* LookUp(const std::string& name) const { auto it = ops_.find(name); return it == ops_.end() ? nullptr : &it->second; }
  const std::vector<std::string>& names() const { return order_; }
  const std::vector<std::string>& duplicates() const { return duplicate_; }

 private:
  std::mutex mu_;
  std::map<std::string, OpDef> ops_;
  std::vector<std::string> order_, duplicate_;
};
struct OpRegistrar {
  OpRegistrar(const OpDefBuilder& b) { OpRegistry::Global().Register(b.def()); }  // NOLINT
};

// ---- kernels -------------------------------------------------------------------------------------
class OpKernel;
class OpKernelConstruction {
 public:
  OpKernelConstruction(const OpDef* def, const AttrMap* attrs) : def_(def) { reader_.attrs = attrs; }
  template <typename T> Status GetAttr(const std::string& name, T* v) const { return reader_.Get(name, v); }
  void SetStatus(const Status& s) { status_.Update(s); }
  const Status& status() const { return status_; }
  const OpDef* op_def() const { return def_; }
  void CtxFailure(const Status& s) { SetStatus(s); }

 private:
  const OpDef* def_;
  AttrReader reader_;
  Status status_;
};

struct TensorValue {
  Tensor* tensor = nullptr;  // ref inputs: the caller's variable; value inputs: the context's copy
  bool is_ref = false;
};

class OpInputList {
 public:
  OpInputList() = default;
  OpInputList(class OpKernelContext* c, int b, int e) : ctx_(c), begin_(b), end_(e) {}
  int size() const { return end_ - begin_; }
  inline const Tensor& operator[](int i) const;

 private:
  class OpKernelContext* ctx_ = nullptr;
  int begin_ = 0, end_ = 0;
};
class OpOutputList {
 public:
  OpOutputList() = default;
  OpOutputList(class OpKernelContext* c, int b, int e) : ctx_(c), begin_(b), end_(e) {}
  int size() const { return end_ - begin_; }
  inline Status allocate(int i, const TensorShape& s, Tensor** out);
  inline void set(int i, const Tensor& t);

 private:
  class OpKernelContext* ctx_ = nullptr;
  int begin_ = 0, end_ = 0;
};

#define TF_MOCK_RETURN(expr) do { ::tensorflow::Status _s = (expr); if (!_s.ok()) return _s; } while (0)
class OpKernelContext {
 public:
  struct Output {
    bool set = false;
    bool is_ref = false;
    Tensor value;             // value outputs
    Tensor* ref = nullptr;    // ref outputs: the forwarded variable
    int forwarded_from = -1;  // input index of a forwarded ref
  };
  OpKernelContext(const OpDef* def, const AttrMap* attrs) : def_(def), attrs_(attrs) {}

  // ---- what the harness fills -----------------------------------------------------
  void AddInput(const Tensor& t) { held_.push_back(std::unique_ptr<Tensor>(new Tensor(t))); inputs_.push_back(TensorValue{held_.back().get(), false}); }
  void AddRefInput(Tensor* variable) { inputs_.push_back(TensorValue{variable, true}); }
  // checks the inputs against the OpDef (types, ref-ness, arity) and sizes the outputs: what the executor guarantees a kernel
  Status Finalize() {
    int at = 0;
    for (const ArgDef& a : def_->inputs) {
      std::vector<DataType> want;
      TF_MOCK_RETURN(ArgTypes(a, &want));
      in_ranges_[a.name] = {at, at + (int)want.size()};
      for (DataType dt : want) {
        if (at >= (int)inputs_.size()) return errors::InvalidArgument("Op ", def_->name, ": input '", a.name, "' is missing");
        const TensorValue& v = inputs_[(size_t)at];
        if (v.is_ref != a.is_ref)
          return errors::InvalidArgument("Op ", def_->name, ": input '", a.name, "' ", a.is_ref ? "must be" : "must not be", " a Ref");
        if (RemoveRefType(v.tensor->dtype()) != dt)
          return errors::InvalidArgument("Op ", def_->name, ": input '", a.name, "' has dtype ", DataTypeString(v.tensor->dtype()),
                                         ", expected ", DataTypeString(dt));
        ++at;
      }
    }
    if (at != (int)inputs_.size()) return errors::InvalidArgument("Op ", def_->name, ": ", inputs_.size(), " inputs given, ", at, " expected");
    at = 0;
    for (const ArgDef& a : def_->outputs) {
      std::vector<DataType> want;
      TF_MOCK_RETURN(ArgTypes(a, &want));
      out_ranges_[a.name] = {at, at + (int)want.size()};
      for (DataType dt : want) { out_types_.push_back(a.is_ref ? MakeRefType(dt) : dt); ++at; }
    }
    outputs_.assign((size_t)at, Output());
    return Status::OK();
  }

  // ---- the kernel-facing API (core/framework/op_kernel.h) ---------------------------
  int num_inputs() const { return (int)inputs_.size(); }
  int num_outputs() const { return (int)outputs_.size(); }
  bool input_is_ref(int i) const { return inputs_[(size_t)i].is_ref; }
  const Tensor& input(int i) {
    TFM_CHECK(i >= 0 && i < num_inputs(), "OpKernelContext::input(" + std::to_string(i) + ")");
    TFM_CHECK(!inputs_[(size_t)i].is_ref, "OpKernelContext::input() on the Ref input " + std::to_string(i) + " (use mutable_input)");
    return *inputs_[(size_t)i].tensor;
  }
  Tensor mutable_input(int i, bool /*lock_held*/) {
    TFM_CHECK(i >= 0 && i < num_inputs(), "OpKernelContext::mutable_input(" + std::to_string(i) + ")");
    TFM_CHECK(inputs_[(size_t)i].is_ref, "OpKernelContext::mutable_input() on the non-Ref input " + std::to_string(i));
    return *inputs_[(size_t)i].tensor;  // a copy that SHARES the variable's buffer
  }
  Status input_list(const std::string& name, OpInputList* list) {
    auto it = in_ranges_.find(name);
    if (it == in_ranges_.end()) return errors::InvalidArgument("Unknown input name: ", name);
    *list = OpInputList(this, it->second.first, it->second.second);
    return Status::OK();
  }
  Status output_list(const std::string& name, OpOutputList* list) {
    auto it = out_ranges_.find(name);
    if (it == out_ranges_.end()) return errors::InvalidArgument("Unknown output name: ", name);
    *list = OpOutputList(this, it->second.first, it->second.second);
    return Status::OK();
  }
  DataType expected_output_dtype(int i) const { return out_types_[(size_t)i]; }
  Status allocate_output(int i, const TensorShape& shape, Tensor** out) {
    TFM_CHECK(i >= 0 && i < num_outputs(), "OpKernelContext::allocate_output(" + std::to_string(i) + ")");
    const DataType dt = out_types_[(size_t)i];
    if (IsRefType(dt)) return errors::Internal("allocate_output on the Ref output ", i);
    std::lock_guard<std::mutex> lk(mu_);
    Output& o = outputs_[(size_t)i];
    o.set = true;
    o.value = Tensor(dt, shape);
    *out = &o.value;
    return Status::OK();
  }
  Status allocate_temp(DataType dt, const TensorShape& shape, Tensor* out) { *out = Tensor(dt, shape); return Status::OK(); }
  void set_output(int i, const Tensor& t) {
    TFM_CHECK(i >= 0 && i < num_outputs(), "OpKernelContext::set_output(" + std::to_string(i) + ")");
    TFM_CHECK(t.dtype() == out_types_[(size_t)i], "set_output: dtype " + DataTypeString(t.dtype()) + " where the OpDef says " +
                                                      DataTypeString(out_types_[(size_t)i]));
    std::lock_guard<std::mutex> lk(mu_);
    Output& o = outputs_[(size_t)i];
    o.set = true;
    o.value = t;  // shares the buffer
  }
  void forward_ref_input_to_ref_output(int in, int out) {
    TFM_CHECK(in >= 0 && in < num_inputs() && inputs_[(size_t)in].is_ref, "forward_ref_input_to_ref_output: input is not a Ref");
    TFM_CHECK(out >= 0 && out < num_outputs() && IsRefType(out_types_[(size_t)out]), "forward_ref_input_to_ref_output: output is not a Ref");
    std::lock_guard<std::mutex> lk(mu_);
    Output& o = outputs_[(size_t)out];
    o.set = true;
    o.is_ref = true;
    o.ref = inputs_[(size_t)in].tensor;
    o.forwarded_from = in;
  }
  void SetStatus(const Status& s) { std::lock_guard<std::mutex> lk(mu_); status_.Update(s); }
  void CtxFailure(const Status& s) { SetStatus(s); }
  Status status() { std::lock_guard<std::mutex> lk(mu_); return status_; }
  const Output& output(int i) const { return outputs_[(size_t)i]; }

 private:
  Status ArgTypes(const ArgDef& a, std::vector<DataType>* out) const {
    if (a.fixed != DT_INVALID) { out->push_back(a.fixed); return Status::OK(); }
    AttrReader r;
    r.attrs = attrs_;
    if (!a.type_attr.empty()) {
      DataType dt;
      TF_MOCK_RETURN(r.Get(a.type_attr, &dt));
      out->push_back(dt);
      return Status::OK();
    }
    return r.Get(a.type_list_attr, out);
  }
  friend class OpInputList;
  friend class OpOutputList;
  const OpDef* def_;
  const AttrMap* attrs_;
  std::vector<std::unique_ptr<Tensor>> held_;
  std::vector<TensorValue> inputs_;
  std::vector<DataType> out_types_;
  std::vector<Output> outputs_;
  std::map<std::string, std::pair<int, int>> in_ranges_, out_ranges_;
  std::mutex mu_;
  Status status_;
};

inline const Tensor& OpInputList::operator[](int i) const {
  TFM_CHECK(i >= 0 && i < size(), "OpInputList[" + std::to_string(i) + "]");
  return ctx_->input(begin_ + i);
}
inline Status OpOutputList::allocate(int i, const TensorShape& s, Tensor** out) {
  TFM_CHECK(i >= 0 && i < size(), "OpOutputList::allocate(" + std::to_string(i) + ")");
  return ctx_->allocate_output(begin_ + i, s, out);
}
inline void OpOutputList::set(int i, const Tensor& t) { ctx_->set_output(begin_ + i, t); }

class OpKernel {
 public:
  explicit OpKernel(OpKernelConstruction* c) : name_(c->op_def() ? c->op_def()->name : "") {}
  virtual ~OpKernel() = default;
  virtual void Compute(OpKernelContext*) = 0;
  virtual bool IsExpensive() { return true; }
  virtual class AsyncOpKernel* AsAsync() { return nullptr; }
  const std::string& name() const { return name_; }
  const std::string& type_string() const { return name_; }

 private:
  std::string name_;
};

class AsyncOpKernel : public OpKernel {
 public:
  using OpKernel::OpKernel;
  typedef std::function<void()> DoneCallback;
  virtual void ComputeAsync(OpKernelContext*, DoneCallback done) = 0;
  AsyncOpKernel* AsAsync() final { return this; }
  void Compute(OpKernelContext*) override { TfmCheckFail("AsyncOpKernel::Compute", "the harness calls ComputeAsync"); }
};

// REGISTER_KERNEL_BUILDER(Name("Op").Device(DEVICE_CPU).TypeConstraint<int32>("T"), Class)
static const char* const DEVICE_CPU = "CPU";
static const char* const DEVICE_GPU = "GPU";
struct KernelDef {
  std::string op, device;
  std::vector<std::pair<std::string, DataType>> constraints;
  std::vector<std::string> host_memory;
};
class KernelDefBuilder {
 public:
  explicit KernelDefBuilder(const char* op) { def_.op = op; }
  KernelDefBuilder& Device(const char* d) { def_.device = d; return *this; }
  template <typename T> KernelDefBuilder& TypeConstraint(const char* attr) { def_.constraints.emplace_back(attr, DataTypeToEnum<T>::v()); return *this; }
  KernelDefBuilder& HostMemory(const char* arg) { def_.host_memory.push_back(arg); return *this; }
  const KernelDef& def() const { return def_; }

 private:
  KernelDef def_;
};
inline KernelDefBuilder Name(const char* op) { return KernelDefBuilder(op); }

struct KernelRegistration {
  KernelDef def;
  std::function<OpKernel*(OpKernelConstruction*)> factory;
};
class KernelRegistry {
 public:
  static KernelRegistry& Global() { static KernelRegistry r; return r; }
  void Register(const KernelDef& d, std::function<OpKernel*(OpKernelConstruction*)> f) {
    std::lock_guard<std::mutex> lk(mu_);
    regs_.push_back(KernelRegistration{d, std::move(f)});
  }
  const std::vector<KernelRegistration>& all() const { return regs_; }
  // FindKernelRegistration: device + every type constraint equal to the node's attr
  Status Find(const std::string& op, const std::string& device, const AttrMap& attrs, const KernelRegistration** out) const {
    const KernelRegistration* hit = nullptr;
    for (const auto& r : regs_) {
      if (r.def.op != op || r.def.device != device) continue;
      bool ok = true;
      for (const auto& c : r.def.constraints) {
        auto it = attrs.find(c.first);
        if (it == attrs.end() || it->second.kind != AttrValue::kType || it->second.type != c.second) { ok = false; break; }
      }
      if (!ok) continue;
      if (hit) return errors::InvalidArgument("Multiple OpKernel registrations match NodeDef '", op, "'");
      hit = &r;
    }
    if (!hit) return errors::NotFound("No registered '", op, "' OpKernel for ", device, " devices compatible with node's attrs");
    *out = hit;
    return Status::OK();
  }

 private:
  std::mutex mu_;
  std::vector<KernelRegistration> regs_;
};
struct KernelRegistrar {
  KernelRegistrar(const KernelDefBuilder& b, std::function<OpKernel*(OpKernelConstruction*)> f) { KernelRegistry::Global().Register(b.def(), std::move(f)); }
};

#define TF_MOCK_CAT2(a, b) a##b
#define TF_MOCK_CAT(a, b) TF_MOCK_CAT2(a, b)
#define REGISTER_OP(name) static ::tensorflow::OpRegistrar TF_MOCK_CAT(tf_mock_op_, __COUNTER__) = ::tensorflow::OpDefBuilder(name)
#define REGISTER_KERNEL_BUILDER(builder, ...)                                            \
  static ::tensorflow::KernelRegistrar TF_MOCK_CAT(tf_mock_kernel_, __COUNTER__)(        \
      builder, [](::tensorflow::OpKernelConstruction* c) -> ::tensorflow::OpKernel* { return new __VA_ARGS__(c); })
#define OP_REQUIRES(ctx, cond, status) do { if (!(cond)) { (ctx)->CtxFailure(status); return; } } while (0)
#define OP_REQUIRES_OK(ctx, expr) do { ::tensorflow::Status _s = (expr); if (!_s.ok()) { (ctx)->CtxFailure(_s); return; } } while (0)
#define OP_REQUIRES_ASYNC(ctx, cond, status, done) do { if (!(cond)) { (ctx)->CtxFailure(status); (done)(); return; } } while (0)
#define OP_REQUIRES_OK_ASYNC(ctx, expr, done) do { ::tensorflow::Status _s = (expr); if (!_s.ok()) { (ctx)->CtxFailure(_s); (done)(); return; } } while (0)
#define TF_RETURN_IF_ERROR(expr) do { ::tensorflow::Status _s = (expr); if (!_s.ok()) return _s; } while (0)
}  // namespace tensorflow
