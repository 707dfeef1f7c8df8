// tfm_driver.cc -- the "executor" side of tests/tf_mock: a C interface (ctypes-friendly) through which
// tests/test_tf_shim_gpu.py looks at what nann_amd/tf_ops/nann_tf_ops.cc REGISTERED (op defs, kernel
// registrations), instantiates kernels from a node's attrs the way TensorFlow's executor does (defaults
// filled from the OpDef, allowed sets and minima enforced, kernel chosen by device + type constraints),
// feeds them tensors that alias the test's numpy buffers, runs Compute / ComputeAsync and reads status
// and outputs.  Test infrastructure for OUR shim only.
#include <chrono>
#include <condition_variable>
#include <thread>

#include "tensorflow/core/framework/op_kernel.h"

using namespace tensorflow;

namespace {

thread_local std::string g_err;
thread_local int g_code = 0;
int Fail(const Status& s) { g_err = s.error_message(); g_code = (int)s.code(); return (int)s.code(); }

// attrs arrive as records "name \x1f type \x1f value" separated by \x1e; list items by \x1d
std::vector<std::string> SplitOn(const std::string& s, char sep) {
  std::vector<std::string> out;
  std::string cur;
  for (char c : s) { if (c == sep) { out.push_back(cur); cur.clear(); } else cur.push_back(c); }
  out.push_back(cur);
  return out;
}

Status ParseAttrs(const char* text, AttrMap* attrs) {
  const std::string t = text ? text : "";
  if (t.empty()) return Status::OK();
  for (const std::string& rec : SplitOn(t, '\x1e')) {
    const std::vector<std::string> f = SplitOn(rec, '\x1f');
    if (f.size() != 3) return errors::InvalidArgument("malformed attr record");
    AttrValue v;
    const std::string &name = f[0], &type = f[1], &val = f[2];
    if (type == "bool") { v.kind = AttrValue::kBool; v.b = val == "1" || val == "true"; }
    else if (type == "int") { v.kind = AttrValue::kInt; v.i = std::strtoll(val.c_str(), nullptr, 10); }
    else if (type == "float") { v.kind = AttrValue::kFloat; v.f = std::strtof(val.c_str(), nullptr); }
    else if (type == "string") { v.kind = AttrValue::kString; v.s = val; }
    else if (type == "type") { v.kind = AttrValue::kType; v.type = (DataType)std::atoi(val.c_str()); }
    else if (type == "shape") {
      v.kind = AttrValue::kShape;
      std::vector<int64> dims;
      if (!val.empty())
        for (const std::string& d : SplitOn(val, ',')) dims.push_back(std::strtoll(d.c_str(), nullptr, 10));
      v.shape = TensorShape(dims);
    } else if (type == "list(string)") { v.kind = AttrValue::kListString; if (!val.empty()) v.list_s = SplitOn(val, '\x1d'); }
    else if (type == "list(type)") {
      v.kind = AttrValue::kListType;
      if (!val.empty())
        for (const std::string& d : SplitOn(val, '\x1d')) v.list_type.push_back((DataType)std::atoi(d.c_str()));
    } else if (type == "list(int)") {
      v.kind = AttrValue::kListInt;
      if (!val.empty())
        for (const std::string& d : SplitOn(val, '\x1d')) v.list_i.push_back(std::strtoll(d.c_str(), nullptr, 10));
    } else return errors::InvalidArgument("unknown attr type '", type, "'");
    (*attrs)[name] = v;
  }
  return Status::OK();
}

// NodeDef validation against the OpDef (core/framework/node_def_util.cc ValidateNodeDef + AddDefaultsToNodeDef)
Status ResolveAttrs(const OpDef& def, AttrMap* attrs) {
  if (!def.error.empty()) return errors::InvalidArgument("OpDef of ", def.name, " did not parse: ", def.error);
  for (const auto& kv : *attrs)
    if (!def.FindAttr(kv.first)) return errors::InvalidArgument("NodeDef mentions attr '", kv.first, "' not in Op<name=", def.name, ">");
  for (const AttrDef& a : def.attrs) {
    auto it = attrs->find(a.name);
    if (it == attrs->end()) {
      if (!a.has_default) return errors::InvalidArgument("NodeDef missing attr '", a.name, "' from Op<name=", def.name, ">");
      (*attrs)[a.name] = a.def;
      continue;
    }
    const AttrValue& v = it->second;
    if (std::string(AttrKindName(v.kind)) != a.type)
      return errors::InvalidArgument("AttrValue of '", a.name, "' has type ", AttrKindName(v.kind), ", the Op says ", a.type);
    if (!a.allowed.empty()) {
      std::vector<DataType> got = v.kind == AttrValue::kType ? std::vector<DataType>{v.type} : v.list_type;
      for (DataType dt : got) {
        bool ok = false;
        for (DataType al : a.allowed) ok = ok || al == dt;
        if (!ok) return errors::InvalidArgument("Value for attr '", a.name, "' of ", DataTypeString(dt), " is not in the list of allowed values");
      }
    }
    if (a.has_min) {
      const int64 have = v.kind == AttrValue::kInt ? v.i
                         : v.kind == AttrValue::kListString ? (int64)v.list_s.size()
                         : v.kind == AttrValue::kListType ? (int64)v.list_type.size() : (int64)v.list_i.size();
      if (have < a.min)
        return errors::InvalidArgument(v.kind == AttrValue::kInt ? "Value for attr '" : "Length for attr '", a.name, "' of ", have,
                                       " must be at least minimum ", a.min);
    }
  }
  return Status::OK();
}

struct Kernel {
  const OpDef* def = nullptr;
  AttrMap attrs;
  std::unique_ptr<OpKernel> kernel;
};

struct Ctx {
  Kernel* k = nullptr;
  std::vector<std::unique_ptr<Tensor>> variables;  // Ref inputs
  std::unique_ptr<OpKernelContext> ctx;
  bool finalized = false;
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  int done_calls = 0;
  bool returned = false;                 // ComputeAsync has returned to its caller
  bool done_before_return = false;       // done() ran before ComputeAsync returned
  std::thread::id caller, done_thread;
  std::string status_msg;
};

std::string Describe(const OpDef& d) {
  std::string s = "Op(" + d.name + ")\n";
  for (const ArgDef& a : d.inputs) s += "Input(" + a.name + ": " + a.spec + ")\n";
  for (const ArgDef& a : d.outputs) s += "Output(" + a.name + ": " + a.spec + ")\n";
  for (const AttrDef& a : d.attrs) s += "Attr(" + a.name + ": " + a.spec + ")\n";
  return s;
}

thread_local std::string g_text;

}  // namespace

extern "C" {

const char* tfm_last_error() { return g_err.c_str(); }
int tfm_last_code() { return g_code; }

// names of the registered ops, '\n'-joined, in registration order
const char* tfm_op_list() {
  g_text.clear();
  for (const std::string& n : OpRegistry::Global().names()) g_text += n + "\n";
  return g_text.c_str();
}
// ops registered twice in this process (TensorFlow refuses the second registration)
const char* tfm_duplicate_ops() {
  g_text.clear();
  for (const std::string& n : OpRegistry::Global().duplicates()) g_text += n + "\n";
  return g_text.c_str();
}
// the op's interface as registered: one line per Input / Output / Attr with the spec text verbatim
const char* tfm_op_def(const char* op) {
  const OpDef* d = OpRegistry::Global().LookUp(op);
  if (!d) { Fail(errors::NotFound("Op type not registered '", op, "'")); return nullptr; }
  g_text = Describe(*d);
  if (!d->error.empty()) g_text += "ERROR(" + d->error + ")\n";
  return g_text.c_str();
}
// kernel registrations: "Op|device|attr=dtype,attr=dtype" per line
const char* tfm_kernel_list() {
  g_text.clear();
  for (const KernelRegistration& r : KernelRegistry::Global().all()) {
    g_text += r.def.op + "|" + r.def.device + "|";
    for (size_t i = 0; i < r.def.constraints.size(); ++i)
      g_text += (i ? "," : "") + r.def.constraints[i].first + "=" + DataTypeString(r.def.constraints[i].second);
    g_text += "\n";
  }
  return g_text.c_str();
}

void* tfm_kernel_new(const char* op, const char* attrs_text) {
  const OpDef* def = OpRegistry::Global().LookUp(op);
  if (!def) { Fail(errors::NotFound("Op type not registered '", op, "'")); return nullptr; }
  std::unique_ptr<Kernel> k(new Kernel());
  k->def = def;
  Status s = ParseAttrs(attrs_text, &k->attrs);
  if (s.ok()) s = ResolveAttrs(*def, &k->attrs);
  const KernelRegistration* reg = nullptr;
  if (s.ok()) s = KernelRegistry::Global().Find(op, DEVICE_CPU, k->attrs, &reg);
  if (!s.ok()) { Fail(s); return nullptr; }
  OpKernelConstruction c(def, &k->attrs);
  k->kernel.reset(reg->factory(&c));
  if (!c.status().ok()) { Fail(c.status()); return nullptr; }  // the kernel is destroyed: what TF does on a failed constructor
  return k.release();
}
void tfm_kernel_delete(void* k) { delete static_cast<Kernel*>(k); }
int tfm_kernel_is_async(void* k) { return static_cast<Kernel*>(k)->kernel->AsAsync() != nullptr; }
int tfm_kernel_is_expensive(void* k) { return static_cast<Kernel*>(k)->kernel->IsExpensive(); }

void* tfm_ctx_new(void* kernel) {
  Ctx* c = new Ctx();
  c->k = static_cast<Kernel*>(kernel);
  c->ctx.reset(new OpKernelContext(c->k->def, &c->k->attrs));
  return c;
}
void tfm_ctx_delete(void* c) { delete static_cast<Ctx*>(c); }

// `data` is BORROWED for the context's lifetime (the test keeps the numpy array): a Ref input's in-place updates land
// in the caller's array
int tfm_ctx_add_input(void* cv, int dtype, int ndim, const int64_t* dims, void* data, int is_ref) {
  Ctx* c = static_cast<Ctx*>(cv);
  std::vector<int64> d;
  for (int i = 0; i < ndim; ++i) d.push_back(dims[i]);
  Tensor t = Tensor::Borrow((DataType)dtype, TensorShape(d), data);
  if (is_ref) {
    c->variables.push_back(std::unique_ptr<Tensor>(new Tensor(t)));
    c->ctx->AddRefInput(c->variables.back().get());
  } else {
    c->ctx->AddInput(t);
  }
  return 0;
}

static int Finish(Ctx* c) {
  const Status s = c->ctx->status();
  c->status_msg = s.error_message();
  return (int)s.code();
}

// starts the kernel: Compute runs to completion here; ComputeAsync is called and this returns as soon as IT returns
int tfm_ctx_start(void* cv) {
  Ctx* c = static_cast<Ctx*>(cv);
  if (!c->finalized) {
    const Status s = c->ctx->Finalize();
    if (!s.ok()) { c->ctx->SetStatus(s); std::lock_guard<std::mutex> lk(c->mu); c->done = true; return Fail(s); }
    c->finalized = true;
  }
  c->caller = std::this_thread::get_id();
  if (AsyncOpKernel* a = c->k->kernel->AsAsync()) {
    a->ComputeAsync(c->ctx.get(), [c] {
      std::lock_guard<std::mutex> lk(c->mu);
      ++c->done_calls;
      c->done = true;
      c->done_thread = std::this_thread::get_id();
      c->done_before_return = !c->returned;
      c->cv.notify_all();
    });
    std::lock_guard<std::mutex> lk(c->mu);
    c->returned = true;
  } else {
    c->k->kernel->Compute(c->ctx.get());
    std::lock_guard<std::mutex> lk(c->mu);
    c->returned = true;
    c->done = true;
    c->done_calls = 1;
    c->done_thread = c->caller;
  }
  return 0;
}
// 1: finished (status via tfm_ctx_status); 0: still running after timeout_ms
int tfm_ctx_wait(void* cv, int timeout_ms) {
  Ctx* c = static_cast<Ctx*>(cv);
  std::unique_lock<std::mutex> lk(c->mu);
  return c->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [c] { return c->done; }) ? 1 : 0;
}
int tfm_ctx_run(void* cv) {
  const int rc = tfm_ctx_start(cv);
  if (rc) return rc;
  if (!tfm_ctx_wait(cv, 120000)) return Fail(errors::DeadlineExceeded("the kernel did not call done() within 120 s"));
  return Finish(static_cast<Ctx*>(cv));
}
int tfm_ctx_status(void* cv) { return Finish(static_cast<Ctx*>(cv)); }
const char* tfm_ctx_status_msg(void* cv) { return static_cast<Ctx*>(cv)->status_msg.c_str(); }
// async evidence: {done() calls, done() ran on another thread than ComputeAsync's caller, done() ran before ComputeAsync returned}
void tfm_ctx_async_info(void* cv, int out[3]) {
  Ctx* c = static_cast<Ctx*>(cv);
  std::lock_guard<std::mutex> lk(c->mu);
  out[0] = c->done_calls;
  out[1] = c->done && c->done_thread != c->caller;
  out[2] = c->done_before_return;
}
int tfm_ctx_num_outputs(void* cv) { return static_cast<Ctx*>(cv)->ctx->num_outputs(); }
// info[4] = {set, dtype (+100 for a Ref output), ndim, forwarded-from input or -1}; dims[8]; *data = the buffer
int tfm_ctx_output(void* cv, int i, int info[4], int64_t dims[8], void** data) {
  Ctx* c = static_cast<Ctx*>(cv);
  if (i < 0 || i >= c->ctx->num_outputs()) return Fail(errors::OutOfRange("output ", i));
  const OpKernelContext::Output& o = c->ctx->output(i);
  info[0] = o.set;
  info[3] = o.forwarded_from;
  const Tensor* t = o.is_ref ? o.ref : &o.value;
  if (!o.set || !t) { info[1] = (int)c->ctx->expected_output_dtype(i); info[2] = 0; *data = nullptr; return 0; }
  info[1] = o.is_ref ? (int)MakeRefType(t->dtype()) : (int)t->dtype();
  info[2] = t->dims();
  if (t->dims() > 8) return Fail(errors::Unimplemented("rank > 8"));
  for (int d = 0; d < t->dims(); ++d) dims[d] = t->dim_size(d);
  *data = const_cast<char*>(t->tensor_data().data());
  return 0;
}

// runs the op's shape function.  ranks[i] = -1: unknown rank; dims (flattened, -1 = unknown).  out: one line per output,
// "?" (unknown rank) or "[d0,d1,...]" with ? for unknown dims
int tfm_infer_shapes(const char* op, const char* attrs_text, int n_in, const int* ranks, const int64_t* dims, char* out, int cap) {
  const OpDef* def = OpRegistry::Global().LookUp(op);
  if (!def) return Fail(errors::NotFound("Op type not registered '", op, "'"));
  AttrMap attrs;
  Status s = ParseAttrs(attrs_text, &attrs);
  if (s.ok()) s = ResolveAttrs(*def, &attrs);
  if (!s.ok()) return Fail(s);
  if (!def->shape_fn) return Fail(errors::InvalidArgument("Op ", op, " has no shape function"));
  std::vector<shape_inference::Shape> in;
  size_t at = 0;
  for (int i = 0; i < n_in; ++i) {
    shape_inference::Shape sh;
    if (ranks[i] >= 0) { sh.known_rank = true; for (int d = 0; d < ranks[i]; ++d) sh.dims.push_back(dims[at++]); }
    in.push_back(sh);
  }
  // outputs: one per output arg (list outputs: as many as the list attr holds)
  int n_out = 0;
  for (const ArgDef& a : def->outputs) {
    if (!a.type_list_attr.empty()) n_out += (int)attrs[a.type_list_attr].list_type.size();
    else ++n_out;
  }
  shape_inference::InferenceContext c(&attrs, in, n_out);
  s = def->shape_fn(&c);
  if (!s.ok()) return Fail(s);
  std::string text;
  for (int i = 0; i < n_out; ++i) {
    shape_inference::ShapeHandle h = c.output(i);
    if (!c.RankKnown(h)) { text += "?\n"; continue; }
    text += "[";
    for (int d = 0; d < c.Rank(h); ++d) {
      const int64 v = c.Value(c.Dim(h, d));
      text += (d ? "," : "") + (v < 0 ? std::string("?") : std::to_string(v));
    }
    text += "]\n";
  }
  if ((int)text.size() + 1 > cap) return Fail(errors::OutOfRange("output buffer too small"));
  std::memcpy(out, text.c_str(), text.size() + 1);
  return 0;
}

}  // extern "C"
