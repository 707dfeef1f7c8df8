// nann_tf_ops.cc -- TensorFlow op-kernel shims that keep NANN's custom-op surface
// (same REGISTER_OP names, input/output names, dtypes and attrs as the reference) and
// run the work on the MI355X through libnann_hip.so's C ABI (include/nann_hip.h).
//
// Drop-in story: build this file against the NANN TensorFlow fork's
// headers in place of
//   tensorflow/core/user_ops/beam_search_op/GroupGather_kernel.cc   (GroupGather, T = int32 | int64)
//   tensorflow/core/user_ops/bitmap_op/bitmap_ops.cc                (BitmapRefDifference, T = int32 | int64)
//   tensorflow/core/user_ops/huge_const_op/huge_const_op.cc         (HugeConst)
//   tensorflow/core/user_ops/blaze_op/blaze_xla_kernel.cc           (BlazeXlaOp)
// or load it with tf.load_op_library (the way bitmap_test.py:11 loads ./bitmap_op.so).
// Graphs produced by NANN_impls/nann/delivery/build_opt_graph.py pin these nodes to
// /CPU:0 (:82,110), so the kernels are registered for DEVICE_CPU with host-memory I/O
// and hop to the GPU internally -- the same trick the reference's CPU-placed BlazeXlaOp
// uses (_blaze_real_device, blaze_predictor.cc:205-257).
//
// Only TensorFlow's public op-kernel API is used; everything device-side happens
// behind the C ABI.  Host code stays C++ inside the op kernel, as in the reference.
#include <cstdint>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/shape_inference.h"

#include "nann_hip.h"

using namespace tensorflow;

namespace nann_tf {

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
      return errors::InvalidArgument(op, ": ", msg);
    case NANN_ERR_IO:
      return errors::NotFound(op, ": ", msg);
    case NANN_ERR_UNSUPPORTED:
      return errors::Unimplemented(op, ": ", msg);
    default:
      return errors::Internal(op, ": ", msg);
  }
}

// RAII device buffer
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  ~DeviceBuffer() { if (ptr_) nann_free(ptr_); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  Status Alloc(int64_t bytes) {
    return ToStatus(nann_malloc(&ptr_, bytes > 0 ? bytes : 1), "nann_malloc");
  }
  Status Upload(const void* host, int64_t bytes) {
    TF_RETURN_IF_ERROR(Alloc(bytes));
    return ToStatus(nann_memcpy(ptr_, host, bytes, /*h2d*/ 0, nullptr), "nann_memcpy");
  }
  Status Download(void* host, int64_t bytes) const {
    TF_RETURN_IF_ERROR(ToStatus(nann_memcpy(host, ptr_, bytes, /*d2h*/ 1, nullptr), "nann_memcpy"));
    return ToStatus(nann_stream_synchronize(nullptr), "nann_stream_synchronize");
  }
  template <typename T> T* as() const { return static_cast<T*>(ptr_); }

 private:
  void* ptr_ = nullptr;
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
  // device copy of [host, host + bytes) if it is (a prefix-aligned whole of) a registered buffer
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
  Status Bind(const void* host, int64_t bytes) {
    ptr_ = HugeConstRegistry::Get().Find(host, bytes);
    if (ptr_) return Status::OK();
    TF_RETURN_IF_ERROR(staged_.Upload(host, bytes));
    ptr_ = staged_.as<void>();
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
    if (n_ret > 0) {
      if (std::is_same<T, int32>::value) {
        OP_REQUIRES_OK(ctx, values->Download(out_values->flat<T>().data(), n_ret * 4));
      } else {
        std::vector<int32_t> host((size_t)n_ret);
        OP_REQUIRES_OK(ctx, values->Download(host.data(), n_ret * 4));
        T* o = out_values->flat<T>().data();
        for (int64_t i = 0; i < n_ret; ++i) o[i] = (T)host[(size_t)i];
      }
    }
    OP_REQUIRES_OK(ctx, splits->Download(out_rs->flat<int64>().data(), n_ret_splits * 8));
  }

 private:
  bool unique_ = false;
};

REGISTER_KERNEL_BUILDER(Name("GroupGather").Device(DEVICE_CPU).TypeConstraint<int32>("T"), GroupGatherHip<int32>);
REGISTER_KERNEL_BUILDER(Name("GroupGather").Device(DEVICE_CPU).TypeConstraint<int64>("T"), GroupGatherHip<int64>);

// ---------------------------------------------------------------------------------
// BitmapRefDifference: same interface as bitmap_ops.cc:150-167
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
    if (std::is_same<T, int32>::value) {
      OP_REQUIRES_OK(ctx, d_v.Upload(values.flat<T>().data(), n * 4));
    } else {
      std::vector<int32_t> narrowed;
      OP_REQUIRES_OK(ctx, NarrowToInt32(reinterpret_cast<const int64*>(values.flat<T>().data()), n, &narrowed,
                                        "idx_next_values"));
      OP_REQUIRES_OK(ctx, d_v.Upload(narrowed.data(), n * 4));
    }
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
    if (n_out > 0) {
      if (std::is_same<T, int32>::value) {
        OP_REQUIRES_OK(ctx, d_out.Download(c_values->flat<T>().data(), n_out * 4));
      } else {
        std::vector<int32_t> host((size_t)n_out);
        OP_REQUIRES_OK(ctx, d_out.Download(host.data(), n_out * 4));
        T* o = c_values->flat<T>().data();
        for (int64_t i = 0; i < n_out; ++i) o[i] = (T)host[(size_t)i];
      }
    }
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
    if (std::is_same<T, int32>::value) {
      OP_REQUIRES_OK(ctx, d_v.Upload(values.flat<T>().data(), n * 4));
    } else {
      std::vector<int32_t> narrowed;
      OP_REQUIRES_OK(ctx, NarrowToInt32(reinterpret_cast<const int64*>(values.flat<T>().data()), n, &narrowed,
                                        "idx_next_values"));
      OP_REQUIRES_OK(ctx, d_v.Upload(narrowed.data(), n * 4));
    }
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
    if (n_out > 0) {
      std::vector<int32_t> host((size_t)n_out);
      OP_REQUIRES_OK(ctx, d_out.Download(host.data(), n_out * 4));
      T* o = c_values->flat<T>().data();
      for (int64_t i = 0; i < n_out; ++i) o[i] = (T)host[(size_t)i];
    }
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
    .SetShapeFn(shape_inference::UnknownShape);

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
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({rows, k}), &value));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, TensorShape({rows, k}), &index));
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
    DataType dtype;
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
    OP_REQUIRES_OK(ctx, ToStatus(nann_memcpy(const_cast<char*>(tensor_.tensor_data().data()), dev_, bytes,
                                             /*d2h*/ 1, nullptr), "HugeConst"));
    OP_REQUIRES_OK(ctx, ToStatus(nann_stream_synchronize(nullptr), "HugeConst"));
    HugeConstRegistry::Get().Add(tensor_.tensor_data().data(), bytes_, dev_);
  }
  ~HugeConstHip() override {
    if (dev_) {
      HugeConstRegistry::Get().Remove(tensor_.tensor_data().data());
      nann_free(dev_);
    }
  }
  void Compute(OpKernelContext* ctx) override { ctx->set_output(0, tensor_); }
  bool IsExpensive() override { return false; }

 private:
  Tensor tensor_;
  void* dev_ = nullptr;
  int64_t bytes_ = 0;
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
// the one output is .../logits f32 [n, 1] (model.py:226-227).  `blaze_option_path` is accepted
// and ignored (XLA warm-up sizes, thread pool and wait_ms have no counterpart: there is no
// compilation step and no nested session to throttle).
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

class BlazeXlaOpHip : public AsyncOpKernel {
 public:
  explicit BlazeXlaOpHip(OpKernelConstruction* ctx) : AsyncOpKernel(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("input_names", &input_names_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("output_names", &output_names_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("graph_def", &model_dir_));
    OP_REQUIRES(ctx, output_names_.size() == 1, errors::InvalidArgument("BlazeXlaOp: one output (logits) expected"));
    for (size_t i = 0; i < input_names_.size(); ++i) {
      if (input_names_[i].find("user_seq_emb") != std::string::npos) user_in_ = (int)i;
      if (input_names_[i].find("item_emb") != std::string::npos) item_in_ = (int)i;
    }
    OP_REQUIRES(ctx, user_in_ >= 0 && item_in_ >= 0,
                errors::InvalidArgument("BlazeXlaOp: input_names must name user_seq_emb and item_emb"));
  }
  ~BlazeXlaOpHip() override { if (model_) nann_model_destroy(model_); }

  void ComputeAsync(OpKernelContext* ctx, DoneCallback done) override {
    OpInputList in;
    OP_REQUIRES_OK_ASYNC(ctx, ctx->input_list("in_tensor", &in), done);
    const Tensor& user = in[user_in_];
    const Tensor& item = in[item_in_];
    OP_REQUIRES_ASYNC(ctx, user.dtype() == DT_HALF && item.dtype() == DT_HALF,
                      errors::InvalidArgument("BlazeXlaOp: float16 inputs expected (build_opt_graph.py:76-92)"), done);
    OP_REQUIRES_ASYNC(ctx, user.dims() == 3 && item.dims() == 2,
                      errors::InvalidArgument("BlazeXlaOp: user_seq_emb [1, L, E] and item_emb [n, d] expected"), done);
    const int64_t n = item.dim_size(0), d = item.dim_size(1);
    const int seq_len = (int)user.dim_size(1);
    {  // the model is loaded on first use: d and L come with the first request
      std::lock_guard<std::mutex> lk(mu_);
      if (!model_) {
        OP_REQUIRES_OK_ASYNC(ctx, ToStatus(nann_model_load(model_dir_.c_str(), (int32_t)d, NANN_F16, seq_len, &model_),
                                           "BlazeXlaOp"), done);
        model_d_ = d;
        model_seq_len_ = seq_len;
      }
    }
    // every later request must have the shapes the model was loaded for: the kernels size their reads from the
    // model, not from the tensors (the reference fails such a request in PadToStatic, blaze_xla_predictor.cc:234-263)
    const int64_t user_e = nann_model_kind(model_) == NANN_MODEL_ATTENTION ? 64 : model_d_;
    OP_REQUIRES_ASYNC(ctx, user.dim_size(0) == 1 && user.dim_size(1) == model_seq_len_ && user.dim_size(2) == user_e,
                      errors::InvalidArgument("BlazeXlaOp: user_seq_emb must be [1, ", model_seq_len_, ", ", user_e,
                                              "] for this model, got ", user.shape().DebugString()), done);
    OP_REQUIRES_ASYNC(ctx, d == model_d_,
                      errors::InvalidArgument("BlazeXlaOp: item_emb must be [n, ", model_d_, "] for this model, got ",
                                              item.shape().DebugString()), done);
    // zero candidates: the reference fails in PadToStatic (blaze_xla_predictor.cc:259-263)
    OP_REQUIRES_ASYNC(ctx, n > 0, errors::Internal("Error when getting input address or size"), done);
    int64_t ws_bytes = 0;
    OP_REQUIRES_OK_ASYNC(ctx, ToStatus(nann_model_workspace_bytes(model_, &ws_bytes), "BlazeXlaOp"), done);
    DeviceInput d_item;  // resident when the rows come straight from a HugeConst, staged otherwise
    DeviceBuffer d_user, d_ws, d_out;
    OP_REQUIRES_OK_ASYNC(ctx, d_user.Upload(user.tensor_data().data(), user.NumElements() * 2), done);
    OP_REQUIRES_OK_ASYNC(ctx, d_item.Bind(item.tensor_data().data(), n * d * 2), done);
    OP_REQUIRES_OK_ASYNC(ctx, d_ws.Alloc(ws_bytes), done);
    OP_REQUIRES_OK_ASYNC(ctx, d_out.Alloc(n * 4), done);
    OP_REQUIRES_OK_ASYNC(ctx, ToStatus(nann_model_forward(model_, d_user.as<void>(), d_item.as<void>(), n,
                                                          d_out.as<float>(), d_ws.as<void>(), nullptr),
                                       "BlazeXlaOp"), done);
    OpOutputList out;
    OP_REQUIRES_OK_ASYNC(ctx, ctx->output_list("out_tensor", &out), done);
    Tensor* logits = nullptr;
    OP_REQUIRES_OK_ASYNC(ctx, out.allocate(0, TensorShape({n, 1}), &logits), done);  // model.py:226-227
    OP_REQUIRES_OK_ASYNC(ctx, d_out.Download(logits->flat<float>().data(), n * 4), done);
    done();
  }

 private:
  std::vector<std::string> input_names_, output_names_;
  std::string model_dir_;
  int user_in_ = -1, item_in_ = -1;
  std::mutex mu_;
  nann_model* model_ = nullptr;
  int64_t model_d_ = 0, model_seq_len_ = 0;  // what the model was loaded for (first request)
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
    const int64_t batch = seq.dim_size(0);
    OP_REQUIRES(ctx, seq.NumElements() == batch * seq_len_ * d_,
                errors::InvalidArgument("comm_seq must be [B, seq_len * emb_dim]"));
    const int32_t* t = topn.flat<int32>().data();
    const int32_t k = t[5];
    DeviceBuffer d_seq, d_q, d_ws, d_ids, d_status;
    OP_REQUIRES_OK(ctx, d_seq.Upload(seq.flat<Eigen::half>().data(), seq.NumElements() * 2));
    OP_REQUIRES_OK(ctx, d_ids.Alloc(batch * k * 8));
    OP_REQUIRES_OK(ctx, d_status.Alloc(batch * 4));
    int64_t ws_bytes = 0;
    if (model_) {  // any model a BlazeXlaOp node could name, the attention + DNN model included
      OP_REQUIRES_OK(ctx, ToStatus(nann_search_model_workspace_bytes(index_, model_, t, batch, &ws_bytes), "NannHnswSearch"));
      OP_REQUIRES_OK(ctx, d_ws.Alloc(ws_bytes));
      OP_REQUIRES_OK(ctx, ToStatus(nann_search_model(index_, model_, d_seq.as<void>(), batch, t, d_ws.as<void>(), ws_bytes,
                                                     d_ids.as<int64_t>(), nullptr, nullptr, d_status.as<int32_t>(),
                                                     nullptr, nullptr),
                                   "NannHnswSearch"));
    } else {
      OP_REQUIRES_OK(ctx, d_q.Alloc(batch * d_ * 4));
      OP_REQUIRES_OK(ctx, ToStatus(nann_user_seq_mean(d_seq.as<void>(), batch, seq_len_, d_, d_q.as<float>(), nullptr),
                                   "NannHnswSearch"));
      OP_REQUIRES_OK(ctx, ToStatus(nann_search_workspace_bytes(index_, t, batch, &ws_bytes), "NannHnswSearch"));
      OP_REQUIRES_OK(ctx, d_ws.Alloc(ws_bytes));
      OP_REQUIRES_OK(ctx, ToStatus(nann_search(index_, scorer_, d_q.as<float>(), batch, t, d_ws.as<void>(), ws_bytes,
                                               d_ids.as<int64_t>(), nullptr, nullptr, d_status.as<int32_t>(),
                                               nullptr, nullptr),
                                   "NannHnswSearch"));
    }
    std::vector<int32_t> status(batch);
    OP_REQUIRES_OK(ctx, d_status.Download(status.data(), batch * 4));
    for (int64_t b = 0; b < batch; ++b)  // a request the reference would have failed
      OP_REQUIRES(ctx, status[b] == NANN_OK,
                  errors::InvalidArgument("request ", b, " failed with nann_status ", status[b]));
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({batch, k}), &out));
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
