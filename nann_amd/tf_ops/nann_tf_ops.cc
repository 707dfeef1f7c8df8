// nann_tf_ops.cc -- TensorFlow op-kernel shims that keep NANN's custom-op surface
// (same REGISTER_OP names, input/output names, dtypes and attrs as the reference) and
// run the work on the MI355X through libnann_hip.so's C ABI (include/nann_hip.h).
//
// Drop-in story: build this file against the NANN TensorFlow fork's
// headers in place of
//   tensorflow/core/user_ops/beam_search_op/GroupGather_kernel.cc   (GroupGather, T = int32 | int64)
//   tensorflow/core/user_ops/bitmap_op/bitmap_ops.cc                (BitmapInit, BitmapDifference, BitmapRefDifference,
//                                                                    BloomFilterDifference; T = int32 | int64)
//   tensorflow/core/user_ops/huge_const_op/huge_const_op.cc         (HugeConst)
//   tensorflow/core/user_ops/blaze_op/blaze_xla_kernel.cc           (BlazeXlaOp)
//   tensorflow/core/user_ops/topk_op/BlazeTopK_kernel.cc            (BlazeTopK)
//   tensorflow/core/user_ops/topk_op/BatchTopKOnRT_kernel.cc        (BatchTopKOnRT)
// -- SIX files: every op registered below is registered by one of them, and an op may be registered once per
// process -- or load it with tf.load_op_library (the way bitmap_test.py:11 loads ./bitmap_op.so) into a TensorFlow
// that does NOT already link those six (INTEGRATION.md 2 has the double-registration hazard).
// Graphs produced by NANN_impls/nann/delivery/build_opt_graph.py pin these nodes to
// /CPU:0 (:82,110), so the kernels are registered for DEVICE_CPU with host-memory I/O
// and hop to the GPU internally -- the same trick the reference's CPU-placed BlazeXlaOp
// uses (_blaze_real_device, blaze_predictor.cc:205-257).
//
// Only TensorFlow's public op-kernel API is used; everything device-side happens
// behind the C ABI.  Host code stays C++ inside the op kernel, as in the reference.
// tests/test_tf_shim_gpu.py builds this file against tests/tf_mock (a functional model of that API; TensorFlow is not
// in the image) and runs every kernel registered here on the GPU.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/shape_inference.h"

#include "nann_hip.h"

using namespace tensorflow;

namespace nann_tf {

// PCIe traffic and HugeConst-registry use of this op library since load (or the last reset): what a host logs to see
// that constants stay resident, and what tests/test_tf_shim_gpu.py asserts ("no re-upload of a HugeConst output")
struct Stats {
  std::atomic<int64_t> h2d_bytes{0}, d2h_bytes{0}, registry_hits{0}, registry_misses{0}, blaze_runs{0}, blaze_rejected{0};
};
static Stats& stats() { static Stats s; return s; }

// maps nann_status to the error class the reference raises at the cited line
static Status ToStatus(int st, const char* op) {
  if (st == NANN_OK) return Status::OK();
  const char* msg = nann_last_error();
  switch (st) {
    case NANN_ERR_INVALID_RAGGED_PARAMS:
    case NANN_ERR_INVALID_RAGGED_INDICES:
    case NANN_ERR_INVALID_RAGGED_INPUT:
    case NANN_ERR_TOPK_K_GT_N:
    case NANN_ERR_INDEX_OUT_OF_RANGE:
    case NANN_ERR_BAD_ARGUMENT:
    case NANN_ERR_TOPK_SCALAR_INPUT:
    case NANN_ERR_DTYPE_MISMATCH:   // huge_const_op.cc:117-147 raise InvalidArgument for both
    case NANN_ERR_SHAPE_MISMATCH:
      return errors::InvalidArgument(op, ": ", msg);
    case NANN_ERR_IO:
      return errors::NotFound(op, ": ", msg);
    case NANN_ERR_UNSUPPORTED:
      return errors::Unimplemented(op, ": ", msg);
    default:
      return errors::Internal(op, ": ", msg);
  }
}

// RAII device buffer; every copy is enqueued on `stream` (NULL = the default stream)
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  ~DeviceBuffer() { if (ptr_) nann_free(ptr_); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  Status Alloc(int64_t bytes) {
    if (ptr_) { nann_free(ptr_); ptr_ = nullptr; cap_ = 0; }
    bytes = bytes > 0 ? bytes : 1;
    TF_RETURN_IF_ERROR(ToStatus(nann_malloc(&ptr_, bytes), "nann_malloc"));
    cap_ = bytes;
    return Status::OK();
  }
  // grow-only (the per-slot buffers of BlazeXlaOp): no hipMalloc once a slot has seen its largest batch
  Status Reserve(int64_t bytes) { return bytes <= cap_ ? Status::OK() : Alloc(bytes + bytes / 4); }
  Status CopyFrom(const void* host, int64_t bytes, nann_stream_t stream = nullptr) {
    if (bytes <= 0) return Status::OK();
    stats().h2d_bytes += bytes;
    return ToStatus(nann_memcpy(ptr_, host, bytes, /*h2d*/ 0, stream), "nann_memcpy");
  }
  Status Upload(const void* host, int64_t bytes, nann_stream_t stream = nullptr) {
    TF_RETURN_IF_ERROR(Alloc(bytes));
    return CopyFrom(host, bytes, stream);
  }
  // device -> host, then waits for the stream: the host tensor is complete on return
  Status Download(void* host, int64_t bytes, nann_stream_t stream = nullptr) const {
    if (bytes > 0) {
      stats().d2h_bytes += bytes;
      TF_RETURN_IF_ERROR(ToStatus(nann_memcpy(host, ptr_, bytes, /*d2h*/ 1, stream), "nann_memcpy"));
    }
    return ToStatus(nann_stream_synchronize(stream), "nann_stream_synchronize");
  }
  template <typename T> T* as() const { return static_cast<T*>(ptr_); }

 private:
  void* ptr_ = nullptr;
  int64_t cap_ = 0;
};

// Constants the graph feeds from HugeConst nodes (CSR values / row_splits, embeddings: hundreds
// of MB) must not cross PCIe per call.  The HugeConst kernel below loads its file into the host
// tensor it outputs (as the reference does, huge_const_op.cc:85-182) AND once into HBM, and
// registers the pair here for its own lifetime; an op that receives a tensor whose buffer is a
// registered HugeConst output uses the resident copy.  Anything else -- placeholders, Consts,
// a re-used allocator address -- is not in the registry and is uploaded per call, so a stale
// device copy cannot be served (the registry is exact, not a (pointer, size) guess).
class HugeConstRegistry {
 public:
  static HugeConstRegistry& Get() { static HugeConstRegistry r; return r; }
  void Add(const void* host, int64_t bytes, void* dev) {
    std::lock_guard<std::mutex> lk(mu_);
    map_[host] = {dev, bytes};
  }
  void Remove(const void* host) {
    std::lock_guard<std::mutex> lk(mu_);
    map_.erase(host);
  }
  // device copy of [host, host + bytes) if it is the whole of a registered buffer
  void* Find(const void* host, int64_t bytes) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = map_.find(host);
    return (it != map_.end() && it->second.second == bytes) ? it->second.first : nullptr;
  }

 private:
  std::mutex mu_;
  std::unordered_map<const void*, std::pair<void*, int64_t>> map_;
};

// a graph input on the device: the resident copy of a HugeConst output, or a per-call upload
class DeviceInput {
 public:
  Status Bind(const void* host, int64_t bytes, nann_stream_t stream = nullptr, DeviceBuffer* staging = nullptr) {
    ptr_ = HugeConstRegistry::Get().Find(host, bytes);
    if (ptr_) { ++stats().registry_hits; return Status::OK(); }
    ++stats().registry_misses;
    DeviceBuffer* b = staging ? staging : &staged_;
    TF_RETURN_IF_ERROR(staging ? b->Reserve(bytes) : b->Alloc(bytes));
    TF_RETURN_IF_ERROR(b->CopyFrom(host, bytes, stream));
    ptr_ = b->as<void>();
    return Status::OK();
  }
  template <typename T> const T* as() const { return static_cast<const T*>(ptr_); }

 private:
  void* ptr_ = nullptr;
  DeviceBuffer staged_;
};

// T = int64 ids (GroupGather_kernel.cc:177-182, bitmap_ops.cc:428-435): the device kernels work on
// int32 ids, so int64 tensors are narrowed on the host (ids beyond int32 cannot index a shard)
static Status NarrowToInt32(const int64* src, int64_t n, std::vector<int32_t>* out, const char* what) {
  out->resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    if (src[i] < std::numeric_limits<int32_t>::min() || src[i] > std::numeric_limits<int32_t>::max())
      return errors::InvalidArgument(what, "[", i, "] = ", src[i], " does not fit the int32 ids of the device kernels");
    (*out)[(size_t)i] = (int32_t)src[i];
  }
  return Status::OK();
}

// ids of a T tensor on the device as int32 (T = int32: as they are; T = int64: narrowed)
template <typename T>
static Status UploadIds(const Tensor& t, DeviceBuffer* dst, const char* what) {
  const int64_t n = t.NumElements();
  if (std::is_same<T, int32>::value) return dst->Upload(t.flat<T>().data(), n * 4);
  std::vector<int32_t> narrowed;
  TF_RETURN_IF_ERROR(NarrowToInt32(reinterpret_cast<const int64*>(t.flat<T>().data()), n, &narrowed, what));
  return dst->Upload(narrowed.data(), n * 4);
}
// int32 ids of the device into a T tensor
template <typename T>
static Status DownloadIds(const DeviceBuffer& src, int64_t n, Tensor* t) {
  if (n <= 0) return Status::OK();
  if (std::is_same<T, int32>::value) return src.Download(t->flat<T>().data(), n * 4);
  std::vector<int32_t> host((size_t)n);
  TF_RETURN_IF_ERROR(src.Download(host.data(), n * 4));
  T* o = t->flat<T>().data();
  for (int64_t i = 0; i < n; ++i) o[i] = (T)host[(size_t)i];
  return Status::OK();
}

// ---------------------------------------------------------------------------------
// GroupGather: same interface as GroupGather_kernel.cc:18-42
REGISTER_OP("GroupGather")
    .Input("params_values: T")
    .Input("params_row_splits: int64")
    .Input("indices_values: int64")
    .Input("indices_row_splits: int64")
    .Output("ret_values: T")
    .Output("ret_row_splits: int64")
    .Attr("T: {int32, int64}")
    .Attr("unique: bool = false")
    .SetShapeFn([](shape_inference::InferenceContext* c) {
      shape_inference::ShapeHandle unused;
      for (int i = 0; i < 4; ++i) TF_RETURN_IF_ERROR(c->WithRank(c->input(i), 1, &unused));
      c->set_output(0, c->MakeShape({c->UnknownDim()}));
      c->set_output(1, c->input(3));
      return Status::OK();
    });

template <typename T>
class GroupGatherHip : public OpKernel {
 public:
  explicit GroupGatherHip(OpKernelConstruction* ctx) : OpKernel(ctx) {
    // unique=true (GroupGather_kernel.cc:91-131): a group's distinct values; the reference's order is its
    // unordered_set's, this kernel's is first-occurrence order -- one of the orders the reference may produce
    OP_REQUIRES_OK(ctx, ctx->GetAttr("unique", &unique_));
  }

  void Compute(OpKernelContext* ctx) override {
    const Tensor& pv = ctx->input(0);
    const Tensor& prs = ctx->input(1);
    const Tensor& iv = ctx->input(2);
    const Tensor& irs = ctx->input(3);
    const int64_t n_pv = pv.NumElements(), n_prs = prs.NumElements();
    const int64_t n_iv = iv.NumElements(), n_irs = irs.NumElements();
    // HugeConst outputs stay resident; everything else (the per-request frontier) is staged
    DeviceInput d_pv, d_prs;
    std::vector<int32_t> narrowed;
    if (std::is_same<T, int32>::value) {
      OP_REQUIRES_OK(ctx, d_pv.Bind(pv.flat<T>().data(), n_pv * 4));
    } else {
      OP_REQUIRES_OK(ctx, NarrowToInt32(reinterpret_cast<const int64*>(pv.flat<T>().data()), n_pv, &narrowed,
                                        "params_values"));
      OP_REQUIRES_OK(ctx, d_pv.Bind(narrowed.data(), n_pv * 4));
    }
    OP_REQUIRES_OK(ctx, d_prs.Bind(prs.flat<int64>().data(), n_prs * 8));
    DeviceBuffer d_iv, d_irs, d_rs, d_off, d_out;
    OP_REQUIRES_OK(ctx, d_iv.Upload(iv.flat<int64>().data(), n_iv * 8));
    OP_REQUIRES_OK(ctx, d_irs.Upload(irs.flat<int64>().data(), n_irs * 8));
    OP_REQUIRES_OK(ctx, d_rs.Alloc((n_irs > 0 ? n_irs : 1) * 8));
    OP_REQUIRES_OK(ctx, d_off.Alloc((n_iv + 1) * 8));
    int64_t n_ret = 0, n_ret_splits = 0;
    int32_t code = 0;
    const int st = nann_group_gather_count(
        d_prs.as<int64_t>(), n_prs, n_pv, d_iv.as<int64_t>(), n_iv, d_irs.as<int64_t>(),
        n_irs, d_rs.as<int64_t>(), d_off.as<int64_t>(), &n_ret, &n_ret_splits, &code, nullptr);
    if (st == NANN_ERR_INVALID_RAGGED_PARAMS) {  // GroupGather_kernel.cc:62-64
      OP_REQUIRES(ctx, false, errors::InvalidArgument("Invalid RaggedTensor input0 params, code: ", code));
    }
    if (st == NANN_ERR_INVALID_RAGGED_INDICES) {  // :65-67
      OP_REQUIRES(ctx, false, errors::InvalidArgument("Invalid RaggedTensor input1 indices, code: ", code));
    }
    OP_REQUIRES_OK(ctx, ToStatus(st, "GroupGather"));
    if (n_ret > 0) {
      OP_REQUIRES_OK(ctx, d_out.Alloc(n_ret * 4));
      OP_REQUIRES_OK(ctx, ToStatus(nann_group_gather_fill(d_pv.as<int32_t>(), d_prs.as<int64_t>(),
                                                          d_iv.as<int64_t>(), n_iv, d_off.as<int64_t>(),
                                                          d_out.as<int32_t>(), nullptr),
                                   "GroupGather"));
    }
    DeviceBuffer d_uniq, d_uniq_rs, d_scratch;
    DeviceBuffer* values = &d_out;
    DeviceBuffer* splits = &d_rs;
    if (unique_ && n_ret > 0 && n_ret_splits > 1) {  // the set of every group of that list (:91-131)
      int64_t scratch_bytes = 0, n_unique = 0;
      OP_REQUIRES_OK(ctx, ToStatus(nann_group_gather_unique_scratch_bytes(n_ret, n_ret_splits, &scratch_bytes), "GroupGather"));
      OP_REQUIRES_OK(ctx, d_scratch.Alloc(scratch_bytes));
      OP_REQUIRES_OK(ctx, d_uniq.Alloc(n_ret * 4));
      OP_REQUIRES_OK(ctx, d_uniq_rs.Alloc(n_ret_splits * 8));
      OP_REQUIRES_OK(ctx, ToStatus(nann_group_gather_unique(d_out.as<int32_t>(), n_ret, d_rs.as<int64_t>(), n_ret_splits,
                                                            d_scratch.as<void>(), d_uniq.as<int32_t>(),
                                                            d_uniq_rs.as<int64_t>(), &n_unique, nullptr),
                                   "GroupGather"));
      n_ret = n_unique;
      values = &d_uniq;
      splits = &d_uniq_rs;
    }
    Tensor* out_values = nullptr;
    Tensor* out_rs = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({n_ret}), &out_values));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, TensorShape({n_ret_splits}), &out_rs));
    OP_REQUIRES_OK(ctx, DownloadIds<T>(*values, n_ret, out_values));
    OP_REQUIRES_OK(ctx, splits->Download(out_rs->flat<int64>().data(), n_ret_splits * 8));
  }

 private:
  bool unique_ = false;
};

REGISTER_KERNEL_BUILDER(Name("GroupGather").Device(DEVICE_CPU).TypeConstraint<int32>("T"), GroupGatherHip<int32>);
REGISTER_KERNEL_BUILDER(Name("GroupGather").Device(DEVICE_CPU).TypeConstraint<int64>("T"), GroupGatherHip<int64>);

// ---------------------------------------------------------------------------------
// BitmapRefDifference: same interface as bitmap_ops.cc:150-167.
// The Ref bitmap lives in HOST memory (the graph's TemporaryVariable on /CPU:0, build_opt_graph.py:115-118), so a call
// carries it over PCIe both ways: N/8 bytes up and down, 125 KB each at 1M items, six calls per query in the
// op-by-op graph (:119-137) -- 1.5 MB per query of plumbing that the fused NannHnswSearch node does not have
// (INTEGRATION.md 3.1).  No device mirror is kept: the graph zeroes the variable with a host Assign between levels
// (:117-118,131) and a mirror keyed by the buffer address would serve stale bits.
REGISTER_OP("BitmapRefDifference")
    .Input("idx_next_values: T")
    .Input("idx_next_row_splits: int64")
    .Input("idx_flag: Ref (int32)")
    .Output("c_values: T")
    .Output("c_row_splits: int64")
    .Output("idx_flag_new: Ref (int32)")
    .Attr("T: {int32, int64}")
    .SetShapeFn([](shape_inference::InferenceContext* c) {
      shape_inference::ShapeHandle unused;
      for (int i = 0; i < 3; ++i) TF_RETURN_IF_ERROR(c->WithRank(c->input(i), 1, &unused));
      c->set_output(0, c->MakeShape({c->UnknownDim()}));
      c->set_output(1, c->input(1));
      c->set_output(2, c->input(2));
      return Status::OK();
    });

template <typename T>
class BitmapRefDifferenceHip : public OpKernel {
 public:
  explicit BitmapRefDifferenceHip(OpKernelConstruction* ctx) : OpKernel(ctx) {}

  void Compute(OpKernelContext* ctx) override {
    const Tensor& values = ctx->input(0);
    const Tensor& row_splits = ctx->input(1);
    Tensor flags = ctx->mutable_input(2, /*lock_held=*/false);  // Ref input, bitmap_ops.cc:179
    const int64_t n = values.NumElements(), n_rs = row_splits.NumElements();
    const int64_t n_words = flags.NumElements();
    DeviceBuffer d_v, d_rs, d_flags, d_out, d_out_rs;
    OP_REQUIRES_OK(ctx, UploadIds<T>(values, &d_v, "idx_next_values"));
    OP_REQUIRES_OK(ctx, d_rs.Upload(row_splits.flat<int64>().data(), n_rs * 8));
    OP_REQUIRES_OK(ctx, d_flags.Upload(flags.flat<int32>().data(), n_words * 4));
    OP_REQUIRES_OK(ctx, d_out.Alloc((n > 0 ? n : 1) * 4));
    OP_REQUIRES_OK(ctx, d_out_rs.Alloc((n_rs > 0 ? n_rs : 1) * 8));
    int64_t n_out = 0, n_out_splits = 0;
    int32_t code = 0;
    const int st = nann_bitmap_ref_difference(d_v.as<int32_t>(), n, d_rs.as<int64_t>(), n_rs,
                                              d_flags.as<int32_t>(), n_words, d_out.as<int32_t>(),
                                              d_out_rs.as<int64_t>(), &n_out, &n_out_splits, &code, nullptr);
    if (st == NANN_ERR_INVALID_RAGGED_INPUT) {  // bitmap_ops.cc:182-184
      OP_REQUIRES(ctx, false, errors::InvalidArgument("Invalid RaggedTensor input0 a, code: ", code));
    }
    OP_REQUIRES_OK(ctx, ToStatus(st, "BitmapRefDifference"));
    Tensor* c_values = nullptr;
    Tensor* c_rs = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({n_out}), &c_values));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, TensorShape({n_out_splits}), &c_rs));
    OP_REQUIRES_OK(ctx, DownloadIds<T>(d_out, n_out, c_values));
    OP_REQUIRES_OK(ctx, d_out_rs.Download(c_rs->flat<int64>().data(), n_out_splits * 8));
    OP_REQUIRES_OK(ctx, d_flags.Download(flags.flat<int32>().data(), n_words * 4));  // in place
    ctx->forward_ref_input_to_ref_output(2, 2);  // bitmap_ops.cc:238
  }
};

REGISTER_KERNEL_BUILDER(Name("BitmapRefDifference").Device(DEVICE_CPU).TypeConstraint<int32>("T"),
                        BitmapRefDifferenceHip<int32>);
REGISTER_KERNEL_BUILDER(Name("BitmapRefDifference").Device(DEVICE_CPU).TypeConstraint<int64>("T"),
                        BitmapRefDifferenceHip<int64>);

// ---------------------------------------------------------------------------------
// BitmapInit: same interface as bitmap_ops.cc:28-43 ("To Deperacate" there; registered by the file this one replaces, so
// it is registered here).  bitmap[length] with the bits of idx set.  An id beyond the bitmap is InvalidArgument here
// (an out-of-bounds write in the reference, :66-72).
REGISTER_OP("BitmapInit")
    .Input("idx: T")
    .Input("length: int32")
    .Output("bitmap: int32")
    .Attr("T: {int32, int64}")
    .SetShapeFn([](shape_inference::InferenceContext* c) {
      shape_inference::ShapeHandle idx;
      TF_RETURN_IF_ERROR(c->WithRank(c->input(0), 1, &idx));
      shape_inference::DimensionHandle words;  // the value of `length` when it is a graph constant
      TF_RETURN_IF_ERROR(c->MakeDimForScalarInput(1, &words));
      c->set_output(0, c->Vector(words));
      return Status::OK();
    });

template <typename T>
class BitmapInitHip : public OpKernel {
 public:
  explicit BitmapInitHip(OpKernelConstruction* ctx) : OpKernel(ctx) {}
  void Compute(OpKernelContext* ctx) override {
    const Tensor& idx = ctx->input(0);
    const int32 length = ctx->input(1).scalar<int32>()();
    const int64_t n = idx.NumElements();
    OP_REQUIRES(ctx, 0 <= length && n <= length,  // bitmap_ops.cc:56-57, the reference's text
                errors::InvalidArgument("require: length >= idx.size() and length >=0 but", "length:", length,
                                        "idx.size():", n));
    Tensor* bitmap = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({length}), &bitmap));
    if (length == 0) return;
    std::memset(bitmap->flat<int32>().data(), 0, (size_t)length * 4);
    if (n == 0) return;
    const int64 row_splits[2] = {0, (int64)n};
    DeviceBuffer d_v, d_rs, d_flags, d_out, d_out_rs;
    OP_REQUIRES_OK(ctx, UploadIds<T>(idx, &d_v, "idx"));
    OP_REQUIRES_OK(ctx, d_rs.Upload(row_splits, 16));
    OP_REQUIRES_OK(ctx, d_flags.Upload(bitmap->flat<int32>().data(), (int64_t)length * 4));
    OP_REQUIRES_OK(ctx, d_out.Alloc(n * 4));
    OP_REQUIRES_OK(ctx, d_out_rs.Alloc(16));
    int64_t n_out = 0, n_out_splits = 0;
    int32_t code = 0;
    OP_REQUIRES_OK(ctx, ToStatus(nann_bitmap_ref_difference(d_v.as<int32_t>(), n, d_rs.as<int64_t>(), 2, d_flags.as<int32_t>(),
                                                            length, d_out.as<int32_t>(), d_out_rs.as<int64_t>(), &n_out,
                                                            &n_out_splits, &code, nullptr),
                                 "BitmapInit"));
    OP_REQUIRES_OK(ctx, d_flags.Download(bitmap->flat<int32>().data(), (int64_t)length * 4));
  }
};
REGISTER_KERNEL_BUILDER(Name("BitmapInit").Device(DEVICE_CPU).TypeConstraint<int32>("T"), BitmapInitHip<int32>);
REGISTER_KERNEL_BUILDER(Name("BitmapInit").Device(DEVICE_CPU).TypeConstraint<int64>("T"), BitmapInitHip<int64>);

// ---------------------------------------------------------------------------------
// BitmapDifference: same interface as bitmap_ops.cc:83-97 -- BitmapRefDifference's value-semantics predecessor over one
// flat list: the bitmap comes in as a tensor and goes out as a NEW tensor (:112-116), the input is left untouched.
REGISTER_OP("BitmapDifference")
    .Input("idx_next: T")
    .Input("idx_flag: int32")
    .Output("idx_next_new: T")
    .Output("idx_flag_new: int32")
    .Attr("T: {int32, int64}")
    .SetShapeFn([](shape_inference::InferenceContext* c) {
      shape_inference::ShapeHandle unused;
      TF_RETURN_IF_ERROR(c->WithRank(c->input(0), 1, &unused));
      TF_RETURN_IF_ERROR(c->WithRank(c->input(1), 1, &unused));
      c->set_output(0, c->MakeShape({c->UnknownDim()}));
      c->set_output(1, c->input(1));
      return Status::OK();
    });

template <typename T>
class BitmapDifferenceHip : public OpKernel {
 public:
  explicit BitmapDifferenceHip(OpKernelConstruction* ctx) : OpKernel(ctx) {}
  void Compute(OpKernelContext* ctx) override {
    const Tensor& idx = ctx->input(0);
    const Tensor& flags = ctx->input(1);
    const int64_t n = idx.NumElements(), n_words = flags.NumElements();
    Tensor* flags_new = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, TensorShape({n_words}), &flags_new));
    if (n_words) std::memcpy(flags_new->flat<int32>().data(), flags.flat<int32>().data(), (size_t)n_words * 4);
    int64_t n_out = 0, n_out_splits = 0;
    DeviceBuffer d_v, d_rs, d_flags, d_out, d_out_rs;
    if (n > 0) {
      const int64 row_splits[2] = {0, (int64)n};
      OP_REQUIRES_OK(ctx, UploadIds<T>(idx, &d_v, "idx_next"));
      OP_REQUIRES_OK(ctx, d_rs.Upload(row_splits, 16));
      OP_REQUIRES_OK(ctx, d_flags.Upload(flags.flat<int32>().data(), n_words * 4));
      OP_REQUIRES_OK(ctx, d_out.Alloc(n * 4));
      OP_REQUIRES_OK(ctx, d_out_rs.Alloc(16));
      int32_t code = 0;
      OP_REQUIRES_OK(ctx, ToStatus(nann_bitmap_ref_difference(d_v.as<int32_t>(), n, d_rs.as<int64_t>(), 2, d_flags.as<int32_t>(),
                                                              n_words, d_out.as<int32_t>(), d_out_rs.as<int64_t>(), &n_out,
                                                              &n_out_splits, &code, nullptr),
                                   "BitmapDifference"));
    }
    Tensor* idx_new = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({n_out}), &idx_new));
    OP_REQUIRES_OK(ctx, DownloadIds<T>(d_out, n_out, idx_new));
    if (n > 0) OP_REQUIRES_OK(ctx, d_flags.Download(flags_new->flat<int32>().data(), n_words * 4));
  }
};
REGISTER_KERNEL_BUILDER(Name("BitmapDifference").Device(DEVICE_CPU).TypeConstraint<int32>("T"), BitmapDifferenceHip<int32>);
REGISTER_KERNEL_BUILDER(Name("BitmapDifference").Device(DEVICE_CPU).TypeConstraint<int64>("T"), BitmapDifferenceHip<int64>);

// ---------------------------------------------------------------------------------
// BloomFilterDifference: same interface as bitmap_ops.cc:264-286 (registered by the reference next to
// BitmapRefDifference; not wired into the serving graph).  T = int32 (int64 narrowed as above).
REGISTER_OP("BloomFilterDifference")
    .Input("idx_next_values: T")
    .Input("idx_next_row_splits: int64")
    .Input("idx_flag: Ref (int32)")
    .Output("c_values: T")
    .Output("c_row_splits: int64")
    .Output("idx_flag_new: Ref (int32)")
    .Attr("bucket: int >= 0 = 0")
    .Attr("bucket_size: int >= 1")
    .Attr("T: {int32, int64}")
    .SetShapeFn([](shape_inference::InferenceContext* c) {
      shape_inference::ShapeHandle unused;
      for (int i = 0; i < 3; ++i) TF_RETURN_IF_ERROR(c->WithRank(c->input(i), 1, &unused));
      c->set_output(0, c->MakeShape({c->UnknownDim()}));
      c->set_output(1, c->input(1));
      c->set_output(2, c->input(2));
      return Status::OK();
    });

template <typename T>
class BloomFilterDifferenceHip : public OpKernel {
 public:
  explicit BloomFilterDifferenceHip(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("bucket", &bucket_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("bucket_size", &bucket_size_));
  }
  void Compute(OpKernelContext* ctx) override {
    const Tensor& values = ctx->input(0);
    const Tensor& row_splits = ctx->input(1);
    Tensor flags = ctx->mutable_input(2, /*lock_held=*/false);
    const int64_t n = values.NumElements(), n_rs = row_splits.NumElements(), n_words = flags.NumElements();
    DeviceBuffer d_v, d_rs, d_flags, d_out, d_out_rs;
    OP_REQUIRES_OK(ctx, UploadIds<T>(values, &d_v, "idx_next_values"));
    OP_REQUIRES_OK(ctx, d_rs.Upload(row_splits.flat<int64>().data(), n_rs * 8));
    OP_REQUIRES_OK(ctx, d_flags.Upload(flags.flat<int32>().data(), n_words * 4));
    OP_REQUIRES_OK(ctx, d_out.Alloc((n > 0 ? n : 1) * 4));
    OP_REQUIRES_OK(ctx, d_out_rs.Alloc((n_rs > 0 ? n_rs : 1) * 8));
    int64_t n_out = 0, n_out_splits = 0;
    int32_t code = 0;
    const int st = nann_bloom_filter_difference(d_v.as<int32_t>(), n, d_rs.as<int64_t>(), n_rs, d_flags.as<int32_t>(),
                                                n_words, bucket_, bucket_size_, d_out.as<int32_t>(),
                                                d_out_rs.as<int64_t>(), &n_out, &n_out_splits, &code, nullptr);
    if (st == NANN_ERR_INVALID_RAGGED_INPUT) {  // bitmap_ops.cc:310-312
      OP_REQUIRES(ctx, false, errors::InvalidArgument("Invalid RaggedTensor input0 a, code: ", code));
    }
    OP_REQUIRES_OK(ctx, ToStatus(st, "BloomFilterDifference"));
    Tensor* c_values = nullptr;
    Tensor* c_rs = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({n_out}), &c_values));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, TensorShape({n_out_splits}), &c_rs));
    OP_REQUIRES_OK(ctx, DownloadIds<T>(d_out, n_out, c_values));
    OP_REQUIRES_OK(ctx, d_out_rs.Download(c_rs->flat<int64>().data(), n_out_splits * 8));
    OP_REQUIRES_OK(ctx, d_flags.Download(flags.flat<int32>().data(), n_words * 4));  // in place
    ctx->forward_ref_input_to_ref_output(2, 2);
  }

 private:
  int64 bucket_ = 0, bucket_size_ = 1;
};

REGISTER_KERNEL_BUILDER(Name("BloomFilterDifference").Device(DEVICE_CPU).TypeConstraint<int32>("T"),
                        BloomFilterDifferenceHip<int32>);
REGISTER_KERNEL_BUILDER(Name("BloomFilterDifference").Device(DEVICE_CPU).TypeConstraint<int64>("T"),
                        BloomFilterDifferenceHip<int64>);

// ---------------------------------------------------------------------------------
// BlazeTopK: same interface as BlazeTopK_kernel.cc:13-26, T = float (the reference also registers half and
// double).  Its tie order is unspecified (std::partial_sort), so TopKV2's order is one of its answers.
REGISTER_OP("BlazeTopK")
    .Input("input: T")
    .Input("k: Tindices")
    .Output("value: T")
    .Output("index: Tindices")
    .Attr("T: {half, float, double}")
    .Attr("Tindices: {int32}")
    .SetShapeFn([](shape_inference::InferenceContext* c) {  // [..., input_len] -> [..., k] for both outputs
      shape_inference::ShapeHandle out;
      TF_RETURN_IF_ERROR(c->ReplaceDim(c->input(0), -1, c->UnknownDim(), &out));
      c->set_output(0, out);
      c->set_output(1, out);
      return Status::OK();
    });

class BlazeTopKHip : public OpKernel {
 public:
  explicit BlazeTopKHip(OpKernelConstruction* ctx) : OpKernel(ctx) {}
  void Compute(OpKernelContext* ctx) override {
    const Tensor& input = ctx->input(0);
    const int32 k = *ctx->input(1).flat<int32>().data();
    OP_REQUIRES(ctx, input.dims() >= 1, errors::InvalidArgument("input must be >= 1-D"));
    const int64_t cols = input.dim_size(input.dims() - 1);
    const int64_t rows = cols ? input.NumElements() / cols : 0;
    OP_REQUIRES(ctx, 0 <= k && k <= cols,  // BlazeTopK_kernel.cc:47-48
                errors::InvalidArgument("require: 0 <= k <= input_len, but", k, " > ", cols));
    Tensor* value = nullptr;
    Tensor* index = nullptr;
    TensorShape out_shape;  // [..., input_len] -> [..., k]
    for (int i = 0; i + 1 < input.dims(); ++i) out_shape.AddDim(input.dim_size(i));
    out_shape.AddDim(k);
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, out_shape, &value));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, out_shape, &index));
    if (k == 0 || rows == 0) return;
    DeviceBuffer d_in, d_v, d_i;
    OP_REQUIRES_OK(ctx, d_in.Upload(input.flat<float>().data(), rows * cols * 4));
    OP_REQUIRES_OK(ctx, d_v.Alloc(rows * k * 4));
    OP_REQUIRES_OK(ctx, d_i.Alloc(rows * k * 4));
    OP_REQUIRES_OK(ctx, ToStatus(nann_topk(d_in.as<float>(), rows, cols, k, d_v.as<float>(), d_i.as<int32_t>(), nullptr),
                                 "BlazeTopK"));
    OP_REQUIRES_OK(ctx, d_v.Download(value->flat<float>().data(), rows * k * 4));
    OP_REQUIRES_OK(ctx, d_i.Download(index->flat<int32>().data(), rows * k * 4));
  }
};

REGISTER_KERNEL_BUILDER(Name("BlazeTopK").Device(DEVICE_CPU).TypeConstraint<float>("T"), BlazeTopKHip);

// ---------------------------------------------------------------------------------
// BatchTopKOnRT: same interface as BatchTopKOnRT_kernel.cc:25-33; T = float (the reference also registers double and
// half).  Per ragged row the min(k, len) best values (`ascending`: the smallest), their ROW-LOCAL int64 positions and the
// output row_splits (:115-121,141-146).  Equal values: std::partial_sort_copy leaves their order unspecified; this
// kernel keeps input order (TopKV2's), one of the reference's answers.  k scalar or one entry per group (:98-108).
// A negative k is InvalidArgument here (the reference computes a negative output size from it).
REGISTER_OP("BatchTopKOnRT")
    .Input("values_in: T")
    .Input("row_splits_in: int64")
    .Input("k: int64")
    .Output("values_out: T")
    .Output("idx_out: int64")
    .Output("row_splits_out: int64")
    .Attr("T: {double, float, half}")
    .Attr("ascending: bool = false")
    .SetShapeFn([](shape_inference::InferenceContext* c) {
      shape_inference::ShapeHandle values, splits, k;
      TF_RETURN_IF_ERROR(c->WithRank(c->input(0), 1, &values));
      TF_RETURN_IF_ERROR(c->WithRank(c->input(1), 1, &splits));
      TF_RETURN_IF_ERROR(c->WithRankAtMost(c->input(2), 1, &k));
      if (c->Rank(k) == 1 && c->ValueKnown(c->Dim(k, 0)) && c->ValueKnown(c->Dim(splits, 0)) &&
          c->Value(c->Dim(k, 0)) != c->Value(c->Dim(splits, 0)) - 1)
        return errors::InvalidArgument("length of k != number of groups: ", c->Value(c->Dim(k, 0)), " != ",
                                       c->Value(c->Dim(splits, 0)), " - 1");
      c->set_output(0, c->MakeShape({c->UnknownDim()}));
      c->set_output(1, c->MakeShape({c->UnknownDim()}));
      c->set_output(2, c->input(1));
      return Status::OK();
    });

class BatchTopKOnRTHip : public OpKernel {
 public:
  explicit BatchTopKOnRTHip(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("ascending", &ascending_));
  }
  void Compute(OpKernelContext* ctx) override {
    const Tensor& values = ctx->input(0);
    const Tensor& row_splits = ctx->input(1);
    const Tensor& k_in = ctx->input(2);
    const int64_t n = values.NumElements(), n_rs = row_splits.NumElements();
    const int64* rs = row_splits.flat<int64>().data();
    const int code = n_rs == 0 ? 1 : rs[0] != 0 ? 2 : rs[n_rs - 1] != n ? 3 : 0;  // ValidateRaggedTensor, :15-22
    OP_REQUIRES(ctx, code == 0, errors::InvalidArgument("Invalid RaggedTensor input, code: ", code));
    const int64_t groups = n_rs - 1;
    if (groups == 0) {  // void inputs: ([], [], [0]), :86-94
      Tensor* t = nullptr;
      OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({0}), &t));
      OP_REQUIRES_OK(ctx, ctx->allocate_output(1, TensorShape({0}), &t));
      OP_REQUIRES_OK(ctx, ctx->allocate_output(2, TensorShape({1}), &t));
      t->flat<int64>().data()[0] = 0;
      return;
    }
    std::vector<int64> k((size_t)groups);
    if (k_in.dims() == 0) {
      std::fill(k.begin(), k.end(), k_in.scalar<int64>()());
    } else {
      OP_REQUIRES(ctx, k_in.NumElements() == groups,  // :104-106
                  errors::InvalidArgument("Size of k vector does NOT match number of groups: ", k_in.NumElements(), "!=", groups));
      std::memcpy(k.data(), k_in.flat<int64>().data(), (size_t)groups * 8);
    }
    Tensor* rs_out_t = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(2, row_splits.shape(), &rs_out_t));
    int64* rs_out = rs_out_t->flat<int64>().data();
    rs_out[0] = 0;
    for (int64_t g = 0; g < groups; ++g) {
      OP_REQUIRES(ctx, rs[g + 1] >= rs[g], errors::InvalidArgument("row_splits_in must be non-decreasing (row ", g, ")"));
      OP_REQUIRES(ctx, k[(size_t)g] >= 0, errors::InvalidArgument("k must be >= 0, got ", k[(size_t)g], " for row ", g));
      rs_out[g + 1] = rs_out[g] + std::min<int64>(rs[g + 1] - rs[g], k[(size_t)g]);
    }
    const int64_t total = rs_out[groups];
    Tensor* values_out = nullptr;
    Tensor* idx_out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({total}), &values_out));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, TensorShape({total}), &idx_out));
    if (total == 0) return;
    const float* vin = values.flat<float>().data();
    DeviceBuffer d_in, d_v, d_i;
    if (ascending_) {  // the k smallest = the k largest of the negated row; values are read back through the positions
      std::vector<float> neg((size_t)n);
      for (int64_t i = 0; i < n; ++i) neg[(size_t)i] = -vin[i];
      OP_REQUIRES_OK(ctx, d_in.Upload(neg.data(), n * 4));
    } else {
      OP_REQUIRES_OK(ctx, d_in.Upload(vin, n * 4));
    }
    OP_REQUIRES_OK(ctx, d_v.Alloc(total * 4));
    OP_REQUIRES_OK(ctx, d_i.Alloc(total * 4));
    for (int64_t g = 0; g < groups; ++g) {
      const int64_t kk = rs_out[g + 1] - rs_out[g];
      if (kk == 0) continue;
      OP_REQUIRES_OK(ctx, ToStatus(nann_topk(d_in.as<float>() + rs[g], 1, rs[g + 1] - rs[g], (int32_t)kk,
                                             d_v.as<float>() + rs_out[g], d_i.as<int32_t>() + rs_out[g], nullptr),
                                   "BatchTopKOnRT"));
    }
    std::vector<int32_t> pos((size_t)total);
    OP_REQUIRES_OK(ctx, d_i.Download(pos.data(), total * 4));
    float* vo = values_out->flat<float>().data();
    int64* io = idx_out->flat<int64>().data();
    for (int64_t g = 0; g < groups; ++g)
      for (int64_t j = rs_out[g]; j < rs_out[g + 1]; ++j) {
        io[j] = pos[(size_t)j];                 // row-local, :146
        vo[j] = vin[rs[g] + pos[(size_t)j]];    // :145
      }
  }

 private:
  bool ascending_ = false;
};
REGISTER_KERNEL_BUILDER(Name("BatchTopKOnRT").Device(DEVICE_CPU).TypeConstraint<float>("T"), BatchTopKOnRTHip);

// ---------------------------------------------------------------------------------
// HugeConst: same interface as huge_const_op.cc:58-70.  The file is read once at kernel
// construction into the host tensor every Compute returns (zero-copy set_output, :184-226) and,
// in the same breath, into HBM; the pair is registered so that the ops above find the resident
// copy.  Validation of dtype / shape against the attrs follows :108-147.
REGISTER_OP("HugeConst")
    .Output("output: dtype")
    .Attr("dtype: type")
    .Attr("shape: shape")
    .Attr("path: string")
    .SetShapeFn([](shape_inference::InferenceContext* c) {
      TensorShape shape_attr;
      TF_RETURN_IF_ERROR(c->GetAttr("shape", &shape_attr));
      shape_inference::ShapeHandle s;
      TF_RETURN_IF_ERROR(c->MakeShapeFromTensorShape(shape_attr, &s));
      c->set_output(0, s);
      return Status::OK();
    });

static int NannDtype(DataType dt) {
  switch (dt) {
    case DT_HALF: return NANN_F16;
    case DT_FLOAT: return NANN_F32;
    case DT_DOUBLE: return NANN_F64;
    case DT_INT32: return NANN_I32;
    case DT_INT64: return NANN_I64;
    default: return -1;
  }
}

class HugeConstHip : public OpKernel {
 public:
  explicit HugeConstHip(OpKernelConstruction* ctx) : OpKernel(ctx) {
    DataType dtype = DT_INVALID;
    TensorShape shape;
    std::string path;
    OP_REQUIRES_OK(ctx, ctx->GetAttr("dtype", &dtype));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("shape", &shape));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("path", &path));
    const int code = NannDtype(dtype);
    OP_REQUIRES(ctx, code >= 0, errors::Unimplemented("Unsupported DataType."));  // huge_const_op.cc:143-146
    std::vector<int64_t> dims;
    for (int i = 0; i < shape.dims(); ++i) dims.push_back(shape.dim_size(i));
    int64_t bytes = 0;
    // no cast: the op checks the file against its attrs, the Python wrapper rewrote the file beforehand
    OP_REQUIRES_OK(ctx, ToStatus(nann_huge_const_load(path.c_str(), code, dims.data(), (int)dims.size(),
                                                      /*allow_cast=*/0, &dev_, &bytes), "HugeConst"));
    tensor_ = Tensor(dtype, shape);
    bytes_ = bytes;
    OP_REQUIRES(ctx, (int64_t)tensor_.TotalBytes() == bytes,
                errors::Internal("HugeConst: ", bytes, " bytes resident for a tensor of ", tensor_.TotalBytes()));
    stats().d2h_bytes += bytes;
    OP_REQUIRES_OK(ctx, ToStatus(nann_memcpy(const_cast<char*>(tensor_.tensor_data().data()), dev_, bytes,
                                             /*d2h*/ 1, nullptr), "HugeConst"));
    OP_REQUIRES_OK(ctx, ToStatus(nann_stream_synchronize(nullptr), "HugeConst"));
    HugeConstRegistry::Get().Add(tensor_.tensor_data().data(), bytes_, dev_);
    registered_ = true;
  }
  ~HugeConstHip() override {
    if (registered_) HugeConstRegistry::Get().Remove(tensor_.tensor_data().data());
    if (dev_) nann_free(dev_);
  }
  void Compute(OpKernelContext* ctx) override { ctx->set_output(0, tensor_); }
  bool IsExpensive() override { return false; }

 private:
  Tensor tensor_;
  void* dev_ = nullptr;
  int64_t bytes_ = 0;
  bool registered_ = false;
};

REGISTER_KERNEL_BUILDER(Name("HugeConst").Device(DEVICE_CPU), HugeConstHip);

// ---------------------------------------------------------------------------------
// BlazeXlaOp: same interface as blaze_xla_kernel.cc:24-33.  The reference runs the frozen
// scoring GraphDef named by `graph_def` in a nested session, padded to warmed-up static batch
// sizes; here `graph_def` names the same FILE -- the frozen GraphDef convert_meta.py:361-398 writes, text or binary,
// read in the reference's order (ReadTextProto, then ReadBinaryProto: blaze_xla_kernel.cc:169-175) by a dependency-free
// reader that pulls the weights out of its Const nodes (csrc/host/nann_graphdef.h, nann_graphdef_text.h) -- or, for the
// scorers that have no frozen graph in the reference (L2, MLP), a weights directory (include/nann_hip.h,
// nann_model_load); nothing of the graph is executed, the weights feed the hand-written kernels, and the batch is
// scored as it comes -- rows are independent, which is all
// PadToStatic / SliceToDynamic rely on (blaze_xla_predictor.cc:227-315).  Inputs are matched by
// `input_names` (constant.py:9-11): .../user_seq_emb f16 [1, L, E] and .../item_emb f16 [n, d];
// the one output is .../logits f32 [n, 1] (model.py:226-227).
//
// Asynchrony and admission control follow blaze_xla_kernel.cc:87-101,194-258:
//   * ComputeAsync returns at once; the upload, kernel and download of a request run on one of BLAZE_THREADS_NUM
//     (env, default 2: :87-93) worker threads, each with a HIP stream and grow-only device buffers of its own, and
//     done() is called from that thread -- the inter-op thread that delivered the request is never blocked.
//   * At most BLAZE_THREADS_NUM requests run; one that arrives while all run WAITS (the reference re-schedules a closure
//     on its pool until a slot frees, :223-236; here it sits in a FIFO).  With `wait_ms` == 0 in the options the FIFO is
//     bounded by DENSE_MAX_WAITING_COUNT (env, default 10: :95-101): beyond it the request fails at once with
//     Internal("waiting pool is full <n>") (:231-233).  With `wait_ms` > 0 (opt_default.conf:1 has 5) the bound is a
//     deadline instead: a request still waiting after wait_ms fails with Internal("blaze wait too long <ns>") (:227-229), one
//     that reaches a worker later than that with DeadlineExceeded("blaze wait too long <ns>") (:243-248).
//   * `blaze_option_path` is parsed as the reference parses it (a text-format BlazeKernelOptions file, else the attr
//     string itself; neither -> Internal "parse proto from ... failed", :156-167) by nann_blaze_options_parse; of its
//     fields wait_ms and run_mode act (SKIP: a [batch, 2] output without a run, :183-188), the XLA / warm-up / grappler
//     ones describe machinery that does not exist here.
REGISTER_OP("BlazeXlaOp")
    .Attr("InT: list({int8,int64,float16,float32,int32})")
    .Attr("OutT: list({int8,int64,float16,float32,int32})")
    .Attr("input_names: list(string) >= 0")
    .Attr("output_names: list(string) >= 0")
    .Attr("graph_def: string")
    .Attr("blaze_option_path: string")
    .Input("in_tensor: InT")
    .Output("out_tensor: OutT")
    .SetShapeFn(shape_inference::UnknownShape);

static int64_t EnvInt(const char* name, int64_t def) {  // ReadInt64FromEnvVar
  const char* v = std::getenv(name);
  if (!v || !*v) return def;
  char* end = nullptr;
  const long long x = std::strtoll(v, &end, 10);
  return (end && *end == 0) ? (int64_t)x : def;
}
static uint64_t NowNanos() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

class BlazeXlaOpHip : public AsyncOpKernel {
 public:
  explicit BlazeXlaOpHip(OpKernelConstruction* ctx)
      : AsyncOpKernel(ctx),
        kRunning_((int)std::max<int64_t>(1, EnvInt("BLAZE_THREADS_NUM", 2))),               // blaze_xla_kernel.cc:87-93
        kMaxWaiting_((int)std::max<int64_t>(0, EnvInt("DENSE_MAX_WAITING_COUNT", 10))) {    // :95-101
    std::string options;
    OP_REQUIRES_OK(ctx, ctx->GetAttr("input_names", &input_names_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("output_names", &output_names_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("graph_def", &model_dir_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("blaze_option_path", &options));
    nann_blaze_options opt = {};
    const int st = nann_blaze_options_parse(options.c_str(), &opt);  // :156-167
    OP_REQUIRES(ctx, st == NANN_OK, errors::Internal("parse proto from ", options, " failed"));
    wait_ns_ = (uint64_t)std::max(0, opt.wait_ms) * 1000000ull;  // :153
    skip_ = opt.run_mode == 2;
    OP_REQUIRES(ctx, output_names_.size() == 1, errors::InvalidArgument("BlazeXlaOp: one output (logits) expected"));
    for (size_t i = 0; i < input_names_.size(); ++i) {
      if (input_names_[i].find("user_seq_emb") != std::string::npos) user_in_ = (int)i;
      if (input_names_[i].find("item_emb") != std::string::npos) item_in_ = (int)i;
    }
    OP_REQUIRES(ctx, user_in_ >= 0 && item_in_ >= 0,
                errors::InvalidArgument("BlazeXlaOp: input_names must name user_seq_emb and item_emb"));
    slots_.resize((size_t)kRunning_);
    for (int i = 0; i < kRunning_; ++i) {
      slots_[(size_t)i].reset(new Slot());
      OP_REQUIRES_OK(ctx, ToStatus(nann_stream_create(&slots_[(size_t)i]->stream), "BlazeXlaOp"));
    }
    for (int i = 0; i < kRunning_; ++i) workers_.emplace_back([this, i] { WorkerLoop(slots_[(size_t)i].get()); });
    if (wait_ns_ > 0) reaper_ = std::thread([this] { ReaperLoop(); });
  }
  ~BlazeXlaOpHip() override {
    std::deque<Request> orphans;
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      orphans.swap(waiting_);
      for (Request& r : ready_) orphans.push_back(std::move(r));
      ready_.clear();
    }
    cv_work_.notify_all();
    cv_reaper_.notify_all();
    for (std::thread& t : workers_) t.join();
    if (reaper_.joinable()) reaper_.join();
    for (Request& r : orphans) {  // the executor never destroys a kernel with steps in flight; be loud if it did
      r.ctx->SetStatus(errors::Internal("BlazeXlaOp destroyed with a request pending"));
      r.done();
    }
    for (auto& s : slots_)
      if (s && s->stream) nann_stream_destroy(s->stream);
    if (model_) nann_model_destroy(model_);
  }

  void ComputeAsync(OpKernelContext* ctx, DoneCallback done) override {
    if (skip_) {  // BlazeKernelOptions::SKIP, blaze_xla_kernel.cc:183-188: a [batch, 2] output, nothing run
      int64 batch = 1;
      for (int i = 0; i < ctx->num_inputs(); ++i) {
        const TensorShape& s = ctx->input(i).shape();
        if (s.dims() != 0 && s.dim_size(0) != 1) { batch = s.dim_size(0); break; }
      }
      Tensor* out = nullptr;
      OP_REQUIRES_OK_ASYNC(ctx, ctx->allocate_output(0, TensorShape({batch, 2}), &out), done);
      done();
      return;
    }
    Request r{ctx, std::move(done), NowNanos()};
    Status reject;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (running_ < kRunning_) {  // a worker is free: blaze_xla_kernel.cc:237-241
        ++running_;
        ready_.push_back(std::move(r));
        cv_work_.notify_one();
        return;
      }
      if (wait_ns_ == 0 && (int)waiting_.size() >= kMaxWaiting_) {  // :231-233
        reject = errors::Internal("waiting pool is full ", waiting_.size());
      } else {  // :234-235 (with wait_ms > 0 the deadline bounds the wait: ReaperLoop, and the check in Run)
        waiting_.push_back(std::move(r));
        cv_reaper_.notify_one();
        return;
      }
    }
    ++stats().blaze_rejected;
    r.ctx->SetStatus(reject);
    r.done();
  }

 private:
  struct Request {
    OpKernelContext* ctx;
    DoneCallback done;
    uint64_t begin;
  };
  struct Slot {  // what one running request owns: a stream and buffers that only grow
    nann_stream_t stream = nullptr;
    DeviceBuffer user, items, ws, out;
  };

  void WorkerLoop(Slot* slot) {
    for (;;) {
      Request r;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_work_.wait(lk, [this] { return stop_ || !ready_.empty(); });
        if (stop_) return;
        r = std::move(ready_.front());
        ready_.pop_front();
      }
      Status st;
      const uint64_t now = NowNanos();
      if (wait_ns_ > 0 && now - r.begin > wait_ns_)  // blaze_xla_kernel.cc:243-248
        st = errors::DeadlineExceeded("blaze wait too long ", now - r.begin);
      else
        st = Run(r.ctx, slot);
      {
        std::lock_guard<std::mutex> lk(mu_);
        --running_;
        if (!waiting_.empty()) {  // the longest waiter takes the freed slot (:237: running_counter_ < kBlazeRunningCount_)
          ready_.push_back(std::move(waiting_.front()));
          waiting_.pop_front();
          ++running_;
          cv_work_.notify_one();
        }
      }
      if (!st.ok()) r.ctx->SetStatus(st);
      r.done();
    }
  }

  // wait_ms > 0: a request that has waited longer than that fails without running (:227-229)
  void ReaperLoop() {
    std::unique_lock<std::mutex> lk(mu_);
    while (!stop_) {
      if (waiting_.empty()) { cv_reaper_.wait(lk); continue; }
      const uint64_t now = NowNanos(), due = waiting_.front().begin + wait_ns_;
      if (now <= due) { cv_reaper_.wait_for(lk, std::chrono::nanoseconds(due - now + 1000)); continue; }
      Request r = std::move(waiting_.front());
      waiting_.pop_front();
      lk.unlock();
      ++stats().blaze_rejected;
      r.ctx->SetStatus(errors::Internal("blaze wait too long ", now - r.begin));
      r.done();
      lk.lock();
    }
  }

  // one request on one slot: BlazeXlaPredictor::Compute's place (blaze_xla_predictor.cc:360-459)
  Status Run(OpKernelContext* ctx, Slot* slot) {
    OpInputList in;
    TF_RETURN_IF_ERROR(ctx->input_list("in_tensor", &in));
    if (in.size() <= std::max(user_in_, item_in_)) return errors::InvalidArgument("BlazeXlaOp: ", in.size(), " inputs for ", input_names_.size(), " input_names");
    const Tensor& user = in[user_in_];
    const Tensor& item = in[item_in_];
    if (user.dtype() != DT_HALF || item.dtype() != DT_HALF)
      return errors::InvalidArgument("BlazeXlaOp: float16 inputs expected (build_opt_graph.py:76-92)");
    if (user.dims() != 3 || item.dims() != 2)
      return errors::InvalidArgument("BlazeXlaOp: user_seq_emb [1, L, E] and item_emb [n, d] expected");
    const int64_t n = item.dim_size(0), d = item.dim_size(1);
    const int seq_len = (int)user.dim_size(1);
    {  // the model is loaded on first use: d and L come with the first request
      std::lock_guard<std::mutex> lk(model_mu_);
      if (!model_) {
        TF_RETURN_IF_ERROR(ToStatus(nann_model_load(model_dir_.c_str(), (int32_t)d, NANN_F16, seq_len, &model_), "BlazeXlaOp"));
        model_d_ = d;
        model_seq_len_ = seq_len;
      }
    }
    // every later request must have the shapes the model was loaded for: the kernels size their reads from the
    // model, not from the tensors (the reference fails such a request in PadToStatic, blaze_xla_predictor.cc:234-263)
    const int64_t user_e = nann_model_kind(model_) == NANN_MODEL_ATTENTION ? 64 : model_d_;
    if (user.dim_size(0) != 1 || user.dim_size(1) != model_seq_len_ || user.dim_size(2) != user_e)
      return errors::InvalidArgument("BlazeXlaOp: user_seq_emb must be [1, ", model_seq_len_, ", ", user_e,
                                     "] for this model, got ", user.shape().DebugString());
    if (d != model_d_)
      return errors::InvalidArgument("BlazeXlaOp: item_emb must be [n, ", model_d_, "] for this model, got ", item.shape().DebugString());
    // zero candidates: the reference fails in PadToStatic (blaze_xla_predictor.cc:259-263)
    if (n <= 0) return errors::Internal("Error when getting input address or size");
    int64_t ws_bytes = 0;
    TF_RETURN_IF_ERROR(ToStatus(nann_model_workspace_bytes(model_, &ws_bytes), "BlazeXlaOp"));
    nann_stream_t s = slot->stream;
    DeviceInput d_item;  // resident when the rows come straight from a HugeConst, staged otherwise
    TF_RETURN_IF_ERROR(slot->user.Reserve(user.NumElements() * 2));
    TF_RETURN_IF_ERROR(slot->user.CopyFrom(user.tensor_data().data(), user.NumElements() * 2, s));
    TF_RETURN_IF_ERROR(d_item.Bind(item.tensor_data().data(), n * d * 2, s, &slot->items));
    TF_RETURN_IF_ERROR(slot->ws.Reserve(ws_bytes));
    TF_RETURN_IF_ERROR(slot->out.Reserve(n * 4));
    TF_RETURN_IF_ERROR(ToStatus(nann_model_forward(model_, slot->user.as<void>(), d_item.as<void>(), n, slot->out.as<float>(),
                                                   slot->ws.as<void>(), s), "BlazeXlaOp"));
    OpOutputList out;
    TF_RETURN_IF_ERROR(ctx->output_list("out_tensor", &out));
    Tensor* logits = nullptr;
    TF_RETURN_IF_ERROR(out.allocate(0, TensorShape({n, 1}), &logits));  // model.py:226-227
    if (logits->dtype() != DT_FLOAT) return errors::InvalidArgument("BlazeXlaOp: OutT must be [float32] (logits)");
    TF_RETURN_IF_ERROR(slot->out.Download(logits->flat<float>().data(), n * 4, s));
    ++stats().blaze_runs;
    return Status::OK();
  }

  const int kRunning_, kMaxWaiting_;
  uint64_t wait_ns_ = 0;
  bool skip_ = false;
  std::vector<std::string> input_names_, output_names_;
  std::string model_dir_;
  int user_in_ = -1, item_in_ = -1;
  std::mutex model_mu_;
  nann_model* model_ = nullptr;
  int64_t model_d_ = 0, model_seq_len_ = 0;  // what the model was loaded for (first request)
  // admission state (blaze_xla_kernel.cc:72-81): running_ = requests on a worker or handed to one
  std::mutex mu_;
  std::condition_variable cv_work_, cv_reaper_;
  std::deque<Request> ready_, waiting_;
  int running_ = 0;
  bool stop_ = false;
  std::vector<std::unique_ptr<Slot>> slots_;
  std::vector<std::thread> workers_;
  std::thread reaper_;
};

REGISTER_KERNEL_BUILDER(Name("BlazeXlaOp").Device(DEVICE_CPU), BlazeXlaOpHip);

// ---------------------------------------------------------------------------------
// NannHnswSearch: the fused schedule as ONE node.  Not a reference op: it replaces the
// ~40-node sub-graph build_model() emits between the `comm_seq`/`level_topn`
// placeholders and `top_k` (build_opt_graph.py:109-149) and is what the serving graph
// should contain on the MI355X.  Paths in the attrs are the same .npy files the
// reference's HugeConst nodes load (build_opt_graph.py:83-90).
REGISTER_OP("NannHnswSearch")
    .Input("comm_seq: float16")   // [B, seq_len * emb_dim]  (reference: [1, 3200])
    .Input("level_topn: int32")   // [6]
    .Output("top_k: int64")       // [B, level_topn[5]]
    .Attr("index_dir: string")
    .Attr("item_embs_dir: string")
    .Attr("seq_len: int = 50")
    .Attr("scorer_dir: string = ''")  // weights directory as for BlazeXlaOp's graph_def (l2 | mlp | attention); '' = L2
    .SetShapeFn(shape_inference::UnknownShape);

class NannHnswSearchHip : public OpKernel {
 public:
  explicit NannHnswSearchHip(OpKernelConstruction* ctx) : OpKernel(ctx) {
    std::string index_dir, embs_dir;
    OP_REQUIRES_OK(ctx, ctx->GetAttr("index_dir", &index_dir));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("item_embs_dir", &embs_dir));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("seq_len", &seq_len_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("scorer_dir", &scorer_dir_));
    OP_REQUIRES(ctx, seq_len_ > 0, errors::InvalidArgument("seq_len must be positive"));
    OP_REQUIRES_OK(ctx, Load(index_dir, embs_dir));
  }
  ~NannHnswSearchHip() override {
    if (index_) nann_index_destroy(index_);
    if (scorer_) nann_scorer_destroy(scorer_);
    if (model_) nann_model_destroy(model_);
    for (void* p : held_) nann_free(p);
  }

  void Compute(OpKernelContext* ctx) override {
    const Tensor& seq = ctx->input(0);
    const Tensor& topn = ctx->input(1);
    OP_REQUIRES(ctx, topn.NumElements() == 6, errors::InvalidArgument("level_topn must have 6 entries"));
    OP_REQUIRES(ctx, seq.dims() == 2, errors::InvalidArgument("comm_seq must be [B, seq_len * emb_dim]"));
    const int64_t batch = seq.dim_size(0);
    const int64_t emb = model_ && nann_model_kind(model_) == NANN_MODEL_ATTENTION ? 64 : d_;
    OP_REQUIRES(ctx, seq.NumElements() == batch * seq_len_ * emb,
                errors::InvalidArgument("comm_seq must be [B, seq_len * emb_dim] = [B, ", seq_len_ * emb, "], got ",
                                        seq.shape().DebugString()));
    const int32_t* t = topn.flat<int32>().data();
    const int32_t k = t[5];
    OP_REQUIRES(ctx, k >= 0, errors::InvalidArgument("level_topn[5] must be >= 0"));
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({batch, k}), &out));
    if (batch == 0 || k == 0) return;
    DeviceBuffer d_seq, d_q, d_ws, d_ids, d_status;
    OP_REQUIRES_OK(ctx, d_seq.Upload(seq.flat<Eigen::half>().data(), seq.NumElements() * 2));
    OP_REQUIRES_OK(ctx, d_ids.Alloc(batch * k * 8));
    OP_REQUIRES_OK(ctx, d_status.Alloc(batch * 4));
    int64_t ws_bytes = 0;
    // the canonical pair of the ABI (v6): nann_search_model_opt / nann_search_opt, uniform level_topn, default options
    if (model_) {  // any model a BlazeXlaOp node could name, the attention + DNN model included
      OP_REQUIRES_OK(ctx, ToStatus(nann_search_model_workspace_bytes(index_, model_, t, batch, &ws_bytes), "NannHnswSearch"));
      OP_REQUIRES_OK(ctx, d_ws.Alloc(ws_bytes));
      OP_REQUIRES_OK(ctx, ToStatus(nann_search_model_opt(index_, model_, d_seq.as<void>(), batch, t, nullptr, d_ws.as<void>(),
                                                         ws_bytes, d_ids.as<int64_t>(), nullptr, nullptr,
                                                         d_status.as<int32_t>(), nullptr, nullptr, nullptr, nullptr),
                                   "NannHnswSearch"));
    } else {
      OP_REQUIRES_OK(ctx, d_q.Alloc(batch * d_ * 4));
      OP_REQUIRES_OK(ctx, ToStatus(nann_user_seq_mean(d_seq.as<void>(), batch, seq_len_, d_, d_q.as<float>(), nullptr),
                                   "NannHnswSearch"));
      OP_REQUIRES_OK(ctx, ToStatus(nann_search_workspace_bytes(index_, t, batch, &ws_bytes), "NannHnswSearch"));
      OP_REQUIRES_OK(ctx, d_ws.Alloc(ws_bytes));
      OP_REQUIRES_OK(ctx, ToStatus(nann_search_opt(index_, scorer_, d_q.as<float>(), batch, t, nullptr, d_ws.as<void>(), ws_bytes,
                                                   d_ids.as<int64_t>(), nullptr, nullptr, d_status.as<int32_t>(), nullptr,
                                                   nullptr, nullptr, nullptr, nullptr),
                                   "NannHnswSearch"));
    }
    std::vector<int32_t> status((size_t)batch);
    OP_REQUIRES_OK(ctx, d_status.Download(status.data(), batch * 4));
    for (int64_t b = 0; b < batch; ++b)  // a request the reference would have failed
      OP_REQUIRES(ctx, status[(size_t)b] == NANN_OK,
                  errors::InvalidArgument("request ", b, " failed with nann_status ", status[(size_t)b]));
    OP_REQUIRES_OK(ctx, d_ids.Download(out->flat<int64>().data(), batch * k * 8));
  }

 private:
  Status LoadNpy(const std::string& path, int dtype, void** dev, int64_t* bytes) {
    // allow_cast: what huge_constant(path, dtype=...) does in build_opt_graph.py:83-90
    const int st = nann_huge_const_load(path.c_str(), dtype, nullptr, 0, /*allow_cast=*/1, dev, bytes);
    if (st == NANN_OK) held_.push_back(*dev);
    return ToStatus(st, "HugeConst");
  }
  Status Load(const std::string& index_dir, const std::string& embs_dir) {
    nann_index_desc d = {};
    int64_t bytes = 0;
    void* p = nullptr;
    TF_RETURN_IF_ERROR(LoadNpy(embs_dir + "/item_ids.npy", NANN_I64, &p, &bytes));
    d.item_ids = static_cast<const int64_t*>(p);
    d.n_items = bytes / 8;
    if (d.n_items <= 0) return errors::InvalidArgument("item_ids.npy is empty");
    TF_RETURN_IF_ERROR(LoadNpy(embs_dir + "/item_embs.npy", NANN_F16, &p, &bytes));
    d.item_embs = p;
    d.d = static_cast<int32_t>(bytes / 2 / d.n_items);
    d.emb_dtype = NANN_F16;
    for (int l = 0; l < 2; ++l) {
      const std::string base = index_dir + "/neighbors_level_" + std::to_string(l);
      TF_RETURN_IF_ERROR(LoadNpy(base + "_values.npy", NANN_I32, &p, &bytes));
      d.nb_values[l] = static_cast<const int32_t*>(p);
      d.nb_nnz[l] = bytes / 4;
      TF_RETURN_IF_ERROR(LoadNpy(base + "_row_splits.npy", NANN_I64, &p, &bytes));
      d.nb_row_splits[l] = static_cast<const int64_t*>(p);
    }
    TF_RETURN_IF_ERROR(LoadNpy(index_dir + "/enter_points.npy", NANN_I32, &p, &bytes));
    d.enter_points = static_cast<const int32_t*>(p);
    d.n_enter = bytes / 4;
    d.on_device = 1;
    d_ = d.d;
    TF_RETURN_IF_ERROR(ToStatus(nann_index_create(&d, &index_), "nann_index_create"));
    nann_scorer_desc s = {};
    s.kind = NANN_SCORER_L2;
    s.d = d.d;
    s.emb_dtype = NANN_F16;
    if (!scorer_dir_.empty())  // l2 | mlp | attention, as for BlazeXlaOp's graph_def
      return ToStatus(nann_model_load(scorer_dir_.c_str(), d.d, NANN_F16, seq_len_, &model_), "scorer");
    return ToStatus(nann_scorer_create(&s, &scorer_), "nann_scorer_create");
  }

  int seq_len_ = 50;
  int d_ = 0;
  std::string scorer_dir_;
  nann_model* model_ = nullptr;
  nann_index* index_ = nullptr;
  nann_scorer* scorer_ = nullptr;
  std::vector<void*> held_;
};

REGISTER_KERNEL_BUILDER(Name("NannHnswSearch").Device(DEVICE_CPU), NannHnswSearchHip);

}  // namespace nann_tf

// counters of this op library for the host's log (and the tests): {bytes host->device, bytes device->host, HugeConst
// registry hits, misses, BlazeXlaOp runs, BlazeXlaOp requests refused by admission control}; reset != 0 zeroes them
extern "C" void nann_tf_ops_stats(int64_t out[6], int reset) {
  nann_tf::Stats& s = nann_tf::stats();
  if (out) {
    out[0] = s.h2d_bytes; out[1] = s.d2h_bytes; out[2] = s.registry_hits; out[3] = s.registry_misses;
    out[4] = s.blaze_runs; out[5] = s.blaze_rejected;
  }
  if (reset) { s.h2d_bytes = 0; s.d2h_bytes = 0; s.registry_hits = 0; s.registry_misses = 0; s.blaze_runs = 0; s.blaze_rejected = 0; }
}
