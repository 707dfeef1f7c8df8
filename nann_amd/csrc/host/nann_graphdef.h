// nann_graphdef.h -- a dependency-free reader of the frozen TensorFlow GraphDef a BlazeXlaOp node names.
//
// The reference's op reads `graph_def` with ReadTextProto / ReadBinaryProto and runs it in a nested
// session (UO/blaze_op/blaze_xla_kernel.cc:169-175, blaze_xla_predictor.cc:360-459).  The file is what
// NANN_impls/nann/delivery/convert_meta.py:361-398 writes: the export graph of Model.forward(training=False)
// (nann/model/model.py:189-233, model_util.py:32-97) with variables frozen to Const nodes, then
// TransformGraph(strip_unused_nodes, merge_duplicate_nodes, fold_constants, sort_by_execution_order), binary
// serialisation (nann/util.py:93-95).  This library has no TensorFlow and runs no graph: it pulls the WEIGHTS
// out of that file -- protobuf wire format, GraphDef -> NodeDef -> attr["value"] -> TensorProto -- and hands
// them to the hand-written scorer kernels (nann_attn_desc).
//
// What survives fold_constants is not the variable list: a variable `V` becomes Const `V` + Identity `V/read`,
// and constant folding replaces whatever is computable from constants by a Const named after the folded OP
// (`<op name>/_<n>__cf__<n>`, common_runtime/constant_folding.cc) -- `V/read`, the Tensordot reshape of a
// kernel, batch norm's `gamma * rsqrt(moving_variance + eps)` -- dropping the originals when nothing else uses
// them.  So tensors are looked up from the CONSUMING op (whose name the Python code fixes: `1_dnn/fc/Tensordot/
// MatMul`, `1_dnn/bn/batchnorm/mul_1`, ...) and the operand is constant-evaluated through whatever chain is
// left (Const | Identity | Reshape | Transpose | Mul | Add | AddV2 | Sub | Rsqrt), with the variable name as
// the fallback.  Both the folded and the merely frozen form load.
//
// Host code, plain C++17, no HIP: included by nann_hip.hip (nann_model_load) and by the CPU test hook in
// host/nann_graphdef_c.cpp.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace nann_gd {

enum { DT_FLOAT = 1, DT_DOUBLE = 2, DT_INT32 = 3, DT_STRING = 7, DT_INT64 = 9, DT_BOOL = 10, DT_BFLOAT16 = 14, DT_HALF = 19 };

struct Tensor {
  int dtype = 0;
  std::vector<int64_t> shape;
  std::vector<float> f;    // float-like payloads (f32 / f64 / f16 / bf16), converted to f32
  std::vector<int64_t> i;  // integer payloads (int32 / int64 / bool): shapes, permutations
  std::vector<std::string> s;  // string_val (DT_STRING)
  // product of the dims; -1 when a dim is negative or the product leaves int64 (found by the wire fuzzer, round 6)
  int64_t count() const {
    int64_t c = 1;
    for (int64_t v : shape) {
      if (v < 0 || (v != 0 && c > INT64_MAX / v)) return -1;
      c *= v;
    }
    return c;
  }
};

// AttrValue (attr_value.proto), every member of its oneof: what the pinning tests compare between the binary and the
// text decode of the same graph.  kind: 's' bytes, 'i' int, 'f' float, 'b' bool, 't' type, 'h' shape, 'T' tensor,
// 'l' list, 'n' func (name only), 'p' placeholder, 0 = empty value.
struct Attr {
  char kind = 0;
  std::string s;
  int64_t i = 0;
  float f = 0.0f;
  bool b = false;
  int type = 0;
  std::vector<int64_t> shape;
  bool unknown_rank = false;
  Tensor tensor;
  std::vector<std::string> ls;
  std::vector<int64_t> li;
  std::vector<float> lf;
  std::vector<int> lb, lt;
  std::vector<std::vector<int64_t>> lshape;
  int n_list_tensors = 0, n_list_funcs = 0;
};

struct Node {
  std::string name, op, device;
  std::vector<std::string> inputs;
  std::map<std::string, Attr> attrs;
  bool has_value = false;  // attr["value"] holds a tensor (a Const's payload: kept ONCE, in the attr -- ADVICE r4)
  const Tensor& value() const { return attrs.at("value").tensor; }
};

struct Graph {
  std::vector<Node> nodes;
  std::unordered_map<std::string, int> by_name;
};

// ---- protobuf wire format ----------------------------------------------------------------------------
struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  bool done() const { return p >= end || !ok; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (p >= end) { ok = false; return 0; }
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    ok = false;
    return 0;
  }
  Cursor sub() {  // a length-delimited field
    const uint64_t n = varint();
    if (!ok || n > (uint64_t)(end - p)) { ok = false; return Cursor{p, p}; }
    Cursor c{p, p + n};
    p += n;
    return c;
  }
  void skip(int wire) {
    switch (wire) {
      case 0: (void)varint(); break;
      case 1: if (end - p < 8) ok = false; else p += 8; break;
      case 2: (void)sub(); break;
      case 5: if (end - p < 4) ok = false; else p += 4; break;
      default: ok = false;  // groups are not used by these messages
    }
  }
  std::string str() { Cursor c = sub(); return std::string(reinterpret_cast<const char*>(c.p), (size_t)(c.end - c.p)); }
};

inline float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, out;
  if (exp == 0) {
    if (man == 0) out = sign;
    else { int e = -1; do { ++e; man <<= 1; } while (!(man & 0x400u)); out = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13); }
  } else if (exp == 31) out = sign | 0x7f800000u | (man << 13);
  else out = sign | ((exp - 15 + 127) << 23) | (man << 13);
  float f; std::memcpy(&f, &out, 4); return f;
}

// TensorProto (tensorflow/core/framework/tensor.proto): dtype = 1, tensor_shape = 2, tensor_content = 4,
// float_val = 5, double_val = 6, int_val = 7, int64_val = 10, half_val = 13 (also carries bfloat16).
// A *_val list shorter than the element count repeats its last value (tensor_util.MakeNdarray).
inline bool parse_tensor(Cursor c, Tensor* t) {
  std::string content;
  std::vector<float> fv;
  std::vector<double> dv;
  std::vector<int64_t> iv;
  std::vector<uint16_t> hv;
  bool unknown_rank = false;
  while (!c.done()) {
    const uint64_t key = c.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (field == 1 && wire == 0) t->dtype = (int)c.varint();
    else if (field == 2 && wire == 2) {  // TensorShapeProto: dim = 2 {size = 1}, unknown_rank = 3
      Cursor s = c.sub();
      while (!s.done()) {
        const uint64_t k2 = s.varint();
        if ((k2 >> 3) == 2 && (k2 & 7) == 2) {
          Cursor d = s.sub();
          int64_t size = 0;
          while (!d.done()) {
            const uint64_t k3 = d.varint();
            if ((k3 >> 3) == 1 && (k3 & 7) == 0) size = (int64_t)d.varint(); else d.skip((int)(k3 & 7));
          }
          if (!d.ok) return false;
          t->shape.push_back(size);
        } else if ((k2 >> 3) == 3 && (k2 & 7) == 0) unknown_rank = s.varint() != 0;
        else s.skip((int)(k2 & 7));
      }
      if (!s.ok) return false;
    } else if (field == 4 && wire == 2) content = c.str();
    else if (field == 5) {  // float_val: packed or not
      if (wire == 2) { Cursor s = c.sub(); while (s.end - s.p >= 4) { float v; std::memcpy(&v, s.p, 4); s.p += 4; fv.push_back(v); } }
      else if (wire == 5) { if (c.end - c.p < 4) return false; float v; std::memcpy(&v, c.p, 4); c.p += 4; fv.push_back(v); }
      else c.skip(wire);
    } else if (field == 6) {
      if (wire == 2) { Cursor s = c.sub(); while (s.end - s.p >= 8) { double v; std::memcpy(&v, s.p, 8); s.p += 8; dv.push_back(v); } }
      else if (wire == 1) { if (c.end - c.p < 8) return false; double v; std::memcpy(&v, c.p, 8); c.p += 8; dv.push_back(v); }
      else c.skip(wire);
    } else if (field == 7 || field == 10) {
      if (wire == 2) { Cursor s = c.sub(); while (!s.done()) iv.push_back((int64_t)s.varint()); if (!s.ok) return false; }
      else if (wire == 0) iv.push_back((int64_t)c.varint());
      else c.skip(wire);
    } else if (field == 13) {
      if (wire == 2) { Cursor s = c.sub(); while (!s.done()) hv.push_back((uint16_t)s.varint()); if (!s.ok) return false; }
      else if (wire == 0) hv.push_back((uint16_t)c.varint());
      else c.skip(wire);
    } else if (field == 8 && wire == 2) t->s.push_back(c.str());  // string_val
    else if (field == 11) {                                       // bool_val
      if (wire == 2) { Cursor s = c.sub(); while (!s.done()) iv.push_back((int64_t)s.varint()); if (!s.ok) return false; }
      else if (wire == 0) iv.push_back((int64_t)c.varint());
      else c.skip(wire);
    } else c.skip(wire);
  }
  if (!c.ok || unknown_rank) return false;
  for (int64_t v : t->shape) if (v < 0) return false;
  const int64_t n = t->count();
  if (n < 0 || n > (int64_t)1 << 31) return false;
  // a *_val list is repeated up to the element count: a 20-byte message may not ask for gigabytes (tensor_content is bounded
  // by the bytes that are there); the reference's model holds 24 576-element tensors at most
  if (content.empty() && n > (int64_t)1 << 24) return false;
  auto fill = [&](auto& dst, const auto& src) {  // repeat the last value
    dst.resize((size_t)n);
    for (int64_t k = 0; k < n; ++k) dst[(size_t)k] = src.empty() ? 0 : src[(size_t)std::min<int64_t>(k, (int64_t)src.size() - 1)];
  };
  switch (t->dtype) {
    case DT_FLOAT:
      if (!content.empty()) {
        if ((int64_t)content.size() != n * 4) return false;
        t->f.resize((size_t)n); std::memcpy(t->f.data(), content.data(), content.size());
      } else fill(t->f, fv);
      return true;
    case DT_DOUBLE:
      if (!content.empty()) {
        if ((int64_t)content.size() != n * 8) return false;
        t->f.resize((size_t)n);
        for (int64_t k = 0; k < n; ++k) { double v; std::memcpy(&v, content.data() + 8 * k, 8); t->f[(size_t)k] = (float)v; }
      } else { std::vector<float> tmp(dv.begin(), dv.end()); fill(t->f, tmp); }
      return true;
    case DT_HALF: case DT_BFLOAT16: {
      std::vector<uint16_t> bits;
      if (!content.empty()) {
        if ((int64_t)content.size() != n * 2) return false;
        bits.resize((size_t)n); std::memcpy(bits.data(), content.data(), content.size());
      } else fill(bits, hv);
      t->f.resize((size_t)n);
      for (int64_t k = 0; k < n; ++k) {
        if (t->dtype == DT_HALF) t->f[(size_t)k] = half_to_float(bits[(size_t)k]);
        else { const uint32_t u = (uint32_t)bits[(size_t)k] << 16; std::memcpy(&t->f[(size_t)k], &u, 4); }
      }
      return true;
    }
    case DT_INT32: case DT_INT64:
      if (!content.empty()) {
        const int es = t->dtype == DT_INT32 ? 4 : 8;
        if ((int64_t)content.size() != n * es) return false;
        t->i.resize((size_t)n);
        for (int64_t k = 0; k < n; ++k) {
          if (es == 4) { int32_t v; std::memcpy(&v, content.data() + 4 * k, 4); t->i[(size_t)k] = v; }
          else { int64_t v; std::memcpy(&v, content.data() + 8 * k, 8); t->i[(size_t)k] = v; }
        }
      } else {
        if (t->dtype == DT_INT32) for (auto& v : iv) v = (int64_t)(int32_t)(uint32_t)v;  // negative int32 arrive sign-extended to 64 bits
        fill(t->i, iv);
      }
      return true;
    case DT_BOOL:
      if (!content.empty()) {
        if ((int64_t)content.size() != n) return false;  // one byte per element, like every other dtype's size check
        t->i.resize((size_t)n);
        for (int64_t k = 0; k < n; ++k) t->i[(size_t)k] = content[(size_t)k] != 0;
      } else fill(t->i, iv);
      return true;
    default:
      return true;  // other dtypes (strings ...): string_val kept as it is, no numeric payload
  }
}

// TensorShapeProto -> dims / unknown_rank
inline bool parse_shape(Cursor s, std::vector<int64_t>* dims, bool* unknown_rank) {
  while (!s.done()) {
    const uint64_t k2 = s.varint();
    if ((k2 >> 3) == 2 && (k2 & 7) == 2) {
      Cursor d = s.sub();
      int64_t size = 0;
      while (!d.done()) {
        const uint64_t k3 = d.varint();
        if ((k3 >> 3) == 1 && (k3 & 7) == 0) size = (int64_t)d.varint(); else d.skip((int)(k3 & 7));
      }
      if (!d.ok) return false;
      dims->push_back(size);
    } else if ((k2 >> 3) == 3 && (k2 & 7) == 0) *unknown_rank = s.varint() != 0;
    else s.skip((int)(k2 & 7));
  }
  return s.ok;
}

// AttrValue: list = 1, s = 2, i = 3, f = 4, b = 5, type = 6, shape = 7, tensor = 8, placeholder = 9, func = 10;
// ListValue: s = 2, i = 3, f = 4, b = 5, type = 6 (packed or not), shape = 7, tensor = 8, func = 9
inline bool parse_attr(Cursor v, Attr* a) {
  auto ints = [](Cursor& c, int wire, auto&& push) {
    if (wire == 2) { Cursor s = c.sub(); while (!s.done()) push((int64_t)s.varint()); return s.ok; }
    if (wire == 0) { push((int64_t)c.varint()); return c.ok; }
    c.skip(wire); return c.ok;
  };
  while (!v.done()) {
    const uint64_t key = v.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (field == 2 && wire == 2) { a->kind = 's'; a->s = v.str(); }
    else if (field == 3 && wire == 0) { a->kind = 'i'; a->i = (int64_t)v.varint(); }
    else if (field == 4 && wire == 5) { a->kind = 'f'; if (v.end - v.p < 4) return false; std::memcpy(&a->f, v.p, 4); v.p += 4; }
    else if (field == 5 && wire == 0) { a->kind = 'b'; a->b = v.varint() != 0; }
    else if (field == 6 && wire == 0) { a->kind = 't'; a->type = (int)v.varint(); }
    else if (field == 7 && wire == 2) { a->kind = 'h'; if (!parse_shape(v.sub(), &a->shape, &a->unknown_rank)) return false; }
    else if (field == 8 && wire == 2) {
      a->kind = 'T';
      Cursor t = v.sub();
      if (!v.ok || !parse_tensor(t, &a->tensor)) return false;
    } else if (field == 9 && wire == 2) { a->kind = 'p'; a->s = v.str(); }
    else if (field == 10 && wire == 2) {  // NameAttrList: name = 1
      a->kind = 'n';
      Cursor f = v.sub();
      while (!f.done()) { const uint64_t k = f.varint(); if ((k >> 3) == 1 && (k & 7) == 2) a->s = f.str(); else f.skip((int)(k & 7)); }
      if (!f.ok) return false;
    } else if (field == 1 && wire == 2) {
      a->kind = 'l';
      Cursor l = v.sub();
      while (!l.done()) {
        const uint64_t k = l.varint();
        const int lf = (int)(k >> 3), lw = (int)(k & 7);
        if (lf == 2 && lw == 2) a->ls.push_back(l.str());
        else if (lf == 3) { if (!ints(l, lw, [&](int64_t x) { a->li.push_back(x); })) return false; }
        else if (lf == 4) {
          if (lw == 2) { Cursor s = l.sub(); while (s.end - s.p >= 4) { float x; std::memcpy(&x, s.p, 4); s.p += 4; a->lf.push_back(x); } }
          else if (lw == 5) { if (l.end - l.p < 4) return false; float x; std::memcpy(&x, l.p, 4); l.p += 4; a->lf.push_back(x); }
          else l.skip(lw);
        }
        else if (lf == 5) { if (!ints(l, lw, [&](int64_t x) { a->lb.push_back(x != 0); })) return false; }
        else if (lf == 6) { if (!ints(l, lw, [&](int64_t x) { a->lt.push_back((int)x); })) return false; }
        else if (lf == 7 && lw == 2) { std::vector<int64_t> d; bool ur = false; if (!parse_shape(l.sub(), &d, &ur)) return false; a->lshape.push_back(d); }
        else if (lf == 8 && lw == 2) { (void)l.sub(); ++a->n_list_tensors; }
        else if (lf == 9 && lw == 2) { (void)l.sub(); ++a->n_list_funcs; }
        else l.skip(lw);
      }
      if (!l.ok) return false;
    } else v.skip(wire);
  }
  return v.ok;
}

// NodeDef (node_def.proto): name = 1, op = 2, input = 3, device = 4, attr = 5 (map entry: key = 1, value = 2)
inline bool parse_node(Cursor c, Node* n) {
  while (!c.done()) {
    const uint64_t key = c.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (field == 1 && wire == 2) n->name = c.str();
    else if (field == 2 && wire == 2) n->op = c.str();
    else if (field == 3 && wire == 2) n->inputs.push_back(c.str());
    else if (field == 4 && wire == 2) n->device = c.str();
    else if (field == 5 && wire == 2) {
      Cursor e = c.sub();
      std::string k;
      Cursor val{nullptr, nullptr};
      bool have_val = false;
      while (!e.done()) {
        const uint64_t k2 = e.varint();
        if ((k2 >> 3) == 1 && (k2 & 7) == 2) k = e.str();
        else if ((k2 >> 3) == 2 && (k2 & 7) == 2) { val = e.sub(); have_val = true; }
        else e.skip((int)(k2 & 7));
      }
      if (!e.ok) return false;
      Attr a;
      if (have_val && !parse_attr(val, &a)) return false;
      if (k == "value" && a.kind == 'T') n->has_value = true;
      n->attrs[k] = std::move(a);
    } else c.skip(wire);
  }
  return c.ok;
}

// GraphDef (graph.proto): node = 1; versions = 4 and library = 2 are skipped.
inline bool parse_graph(const uint8_t* data, size_t n, Graph* g, std::string* err) {
  Cursor c{data, data + n};
  while (!c.done()) {
    const uint64_t key = c.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (field == 1 && wire == 2) {
      Cursor nc = c.sub();
      if (!c.ok) break;
      Node node;
      if (!parse_node(nc, &node)) { *err = "malformed NodeDef (after " + std::to_string(g->nodes.size()) + " nodes)"; return false; }
      g->by_name.emplace(node.name, (int)g->nodes.size());
      g->nodes.push_back(std::move(node));
    } else c.skip(wire);
  }
  if (!c.ok) { *err = "not a binary GraphDef (protobuf wire format error)"; return false; }
  if (g->nodes.empty()) { *err = "GraphDef holds no nodes"; return false; }
  return true;
}

// SavedModel (saved_model.proto): meta_graphs = 2; MetaGraphDef (meta_graph.proto): graph_def = 2.  The graph of the
// first meta graph.  (BlazeXlaOp itself only reads GraphDef files; this is for the TensorFlow-written fixtures the
// fork holds as SavedModels -- tests/test_tf_written_protos.py.)
inline bool parse_saved_model(const uint8_t* data, size_t n, Graph* g, std::string* err) {
  Cursor c{data, data + n};
  while (!c.done()) {
    const uint64_t key = c.varint();
    if ((key >> 3) == 2 && (key & 7) == 2) {
      Cursor mg = c.sub();
      if (!c.ok) break;
      while (!mg.done()) {
        const uint64_t k2 = mg.varint();
        if ((k2 >> 3) == 2 && (k2 & 7) == 2) {
          Cursor gd = mg.sub();
          if (!mg.ok) break;
          return parse_graph(gd.p, (size_t)(gd.end - gd.p), g, err);
        }
        mg.skip((int)(k2 & 7));
      }
      if (!mg.ok) break;
    } else c.skip((int)(key & 7));
  }
  *err = "not a binary SavedModel with a graph_def";
  return false;
}

// ---- constant evaluation of an operand ---------------------------------------------------------------
inline std::string input_node_name(const std::string& in) {  // "^ctrl", "name:0" -> name
  size_t b = (!in.empty() && in[0] == '^') ? 1 : 0;
  const size_t colon = in.rfind(':');
  size_t e = in.size();
  if (colon != std::string::npos && colon > b) {
    bool digits = colon + 1 < in.size();
    for (size_t k = colon + 1; k < in.size(); ++k) digits &= in[k] >= '0' && in[k] <= '9';
    if (digits) e = colon;
  }
  return in.substr(b, e - b);
}

inline bool broadcast2(const Tensor& a, const Tensor& b, Tensor* out, float (*fn)(float, float)) {
  // the shapes these graphs combine: equal, or one side a scalar / single element, or a vector against the
  // last dimension
  const int64_t na = (int64_t)a.f.size(), nb = (int64_t)b.f.size();
  if (na == 0 || nb == 0) return false;
  const Tensor& big = na >= nb ? a : b;
  const int64_t n = std::max(na, nb), small = std::min(na, nb);
  if (n % small) return false;
  if (small != 1 && small != n && (big.shape.empty() || big.shape.back() != small)) return false;
  out->dtype = DT_FLOAT;
  out->shape = big.shape;
  out->f.resize((size_t)n);
  for (int64_t k = 0; k < n; ++k) out->f[(size_t)k] = fn(a.f[(size_t)(na == n ? k : k % na)], b.f[(size_t)(nb == n ? k : k % nb)]);
  return true;
}

inline bool eval(const Graph& g, const std::string& input, Tensor* out, int depth = 0) {
  if (depth > 32) return false;
  const auto it = g.by_name.find(input_node_name(input));
  if (it == g.by_name.end()) return false;
  const Node& n = g.nodes[(size_t)it->second];
  auto data_inputs = [&]() { std::vector<std::string> v; for (const auto& s : n.inputs) if (s.empty() || s[0] != '^') v.push_back(s); return v; };
  if (n.op == "Const") { if (!n.has_value) return false; *out = n.value(); return true; }
  const std::vector<std::string> in = data_inputs();
  if (n.op == "Identity" || n.op == "StopGradient" || n.op == "Snapshot") return in.size() >= 1 && eval(g, in[0], out, depth + 1);
  if (n.op == "Cast") {  // the payload is f32 already; integer -> float casts do not occur on these operands
    return in.size() >= 1 && eval(g, in[0], out, depth + 1) && !out->f.empty();
  }
  if (n.op == "Reshape") {
    Tensor shp;
    if (in.size() < 2 || !eval(g, in[0], out, depth + 1) || !eval(g, in[1], &shp, depth + 1) || shp.i.empty()) return false;
    const int64_t total = (int64_t)std::max(out->f.size(), out->i.size());
    int64_t known = 1, wild = -1;
    for (size_t k = 0; k < shp.i.size(); ++k) { if (shp.i[k] < 0) wild = (int64_t)k; else known *= shp.i[k]; }
    out->shape.assign(shp.i.begin(), shp.i.end());
    if (wild >= 0) { if (known == 0 || total % known) return false; out->shape[(size_t)wild] = total / known; }
    return out->count() == total;
  }
  if (n.op == "Transpose") {
    Tensor src, perm;
    if (in.size() < 2 || !eval(g, in[0], &src, depth + 1) || !eval(g, in[1], &perm, depth + 1)) return false;
    bool identity = perm.i.size() == src.shape.size();
    for (size_t k = 0; k < perm.i.size(); ++k) identity &= perm.i[k] == (int64_t)k;
    if (identity) { *out = src; return true; }
    if (src.shape.size() == 2 && perm.i.size() == 2 && perm.i[0] == 1 && perm.i[1] == 0 && !src.f.empty()) {
      const int64_t r = src.shape[0], c = src.shape[1];
      out->dtype = src.dtype; out->shape = {c, r}; out->f.resize(src.f.size());
      for (int64_t a = 0; a < r; ++a) for (int64_t b = 0; b < c; ++b) out->f[(size_t)(b * r + a)] = src.f[(size_t)(a * c + b)];
      return true;
    }
    return false;
  }
  if (n.op == "Rsqrt") {
    if (in.size() < 1 || !eval(g, in[0], out, depth + 1) || out->f.empty()) return false;
    for (float& v : out->f) v = 1.0f / std::sqrt(v);
    return true;
  }
  if (n.op == "Mul" || n.op == "Add" || n.op == "AddV2" || n.op == "Sub") {
    Tensor a, b;
    if (in.size() < 2 || !eval(g, in[0], &a, depth + 1) || !eval(g, in[1], &b, depth + 1)) return false;
    if (n.op == "Mul") return broadcast2(a, b, out, [](float x, float y) { return x * y; });
    if (n.op == "Sub") return broadcast2(a, b, out, [](float x, float y) { return x - y; });
    return broadcast2(a, b, out, [](float x, float y) { return x + y; });
  }
  return false;  // Placeholder, MatMul, ...: not a constant
}

// node names end with `tail` at a scope boundary ("1_dnn/fc/BiasAdd" matches "tower/1_dnn/fc/BiasAdd", not
// "11_dnn/fc/BiasAdd"); of several (replicas the export kept), the shortest name wins
inline const Node* find_by_tail(const Graph& g, const std::string& tail) {
  const Node* best = nullptr;
  for (const Node& n : g.nodes) {
    if (n.name.size() < tail.size() || n.name.compare(n.name.size() - tail.size(), tail.size(), tail) != 0) continue;
    if (n.name.size() > tail.size() && n.name[n.name.size() - tail.size() - 1] != '/') continue;
    if (!best || n.name.size() < best->name.size()) best = &n;
  }
  return best;
}

// operand `which` (0 | 1; -1 = whichever of the two evaluates to a constant) of the op named `<tail>`
inline bool operand_of(const Graph& g, const std::string& op_tail, int which, Tensor* out) {
  const Node* n = find_by_tail(g, op_tail);
  if (!n) return false;
  std::vector<std::string> in;
  for (const auto& s : n->inputs) if (s.empty() || s[0] != '^') in.push_back(s);
  if (which >= 0) return (int)in.size() > which && eval(g, in[(size_t)which], out);
  for (size_t k = 0; k < in.size() && k < 2; ++k) { Tensor t; if (eval(g, in[k], &t) && !t.f.empty()) { *out = t; return true; } }
  return false;
}

// a variable by its own name: Const `V`, or what folding made of `V/read`
inline bool variable(const Graph& g, const std::string& var, Tensor* out) {
  if (const Node* n = find_by_tail(g, var)) if (n->op == "Const" && n->has_value) { *out = n->value(); return true; }
  const std::string pre = var + "/read";
  for (const Node& n : g.nodes) {
    if (n.op != "Const" || !n.has_value) continue;
    const size_t at = n.name.find(pre);
    if (at == std::string::npos || (at > 0 && n.name[at - 1] != '/')) continue;
    const size_t after = at + pre.size();
    if (after == n.name.size() || n.name[after] == '/') { *out = n.value(); return true; }
  }
  return false;
}

// ---- the reference's scorer model ---------------------------------------------------------------------
struct AttnWeights {
  std::vector<float> wq1, bq1, aq, wq2, bq2, wk1, bk1, ak, wk2, bk2;
  std::vector<float> w[4], b[3], bn_scale[3], bn_shift[3], alpha[3];
  int d = 0;  // item embedding dim, from the shapes
  int e = 0;  // user sequence embedding dim
};

inline bool take(const Tensor& t, int64_t expect, std::vector<float>* dst, const std::string& what, std::string* err) {
  if ((int64_t)t.f.size() != expect) {
    *err = what + ": " + std::to_string(t.f.size()) + " values, expected " + std::to_string(expect);
    return false;
  }
  *dst = t.f;
  return true;
}

// dense layer `scope` (tf.layers.dense: kernel [in, out] as MatMul's second operand, bias as BiasAdd's)
inline bool dense_layer(const Graph& g, const std::string& scope, Tensor* kernel, Tensor* bias, std::string* err) {
  if (!operand_of(g, scope + "/Tensordot/MatMul", 1, kernel) && !operand_of(g, scope + "/MatMul", 1, kernel) &&
      !variable(g, scope + "/kernel", kernel)) {
    *err = "no constant kernel for dense layer '" + scope + "' (looked for " + scope + "/Tensordot/MatMul, " + scope +
           "/MatMul, " + scope + "/kernel)";
    return false;
  }
  if (kernel->shape.size() != 2 || kernel->f.empty()) { *err = "kernel of '" + scope + "' is not a float matrix"; return false; }
  if (bias && !operand_of(g, scope + "/BiasAdd", 1, bias) && !variable(g, scope + "/bias", bias)) {
    *err = "no constant bias for dense layer '" + scope + "'";
    return false;
  }
  return true;
}

// Model.forward (model.py:189-233): nonlinear_attention (model_util.py:70-97: dense, dense_1 on the candidate
// side, dense_2, dense_3 on the sequence side, PReLU slopes prelu_q / prelu_k) then 1_dnn .. 4_dnn (DNN class,
// model_util.py:32-67: fc -> bn (inference form) -> prelu; 4_dnn: fc without bias).
inline bool extract_attention(const Graph& g, AttnWeights* w, std::string* err) {
  const std::string A = "nonlinear_attention";
  Tensor k, b, t;
  if (!dense_layer(g, A + "/dense", &k, &b, err)) return false;
  w->d = (int)k.shape[0];
  const int64_t hq = k.shape[1];
  if (!take(k, (int64_t)w->d * hq, &w->wq1, "nonlinear_attention/dense kernel", err) || !take(b, hq, &w->bq1, "nonlinear_attention/dense bias", err)) return false;
  if (!dense_layer(g, A + "/dense_1", &k, &b, err)) return false;
  const int64_t hq2 = k.shape[1];
  if (k.shape[0] != hq) { *err = "nonlinear_attention/dense_1 kernel does not follow dense"; return false; }
  w->wq2 = k.f; w->bq2 = b.f;
  if (!dense_layer(g, A + "/dense_2", &k, &b, err)) return false;
  w->e = (int)k.shape[0];
  if (k.shape[1] != hq) { *err = "nonlinear_attention/dense_2 width differs from dense"; return false; }
  w->wk1 = k.f; w->bk1 = b.f;
  if (!dense_layer(g, A + "/dense_3", &k, &b, err)) return false;
  if (k.shape[0] != hq || k.shape[1] != hq2) { *err = "nonlinear_attention/dense_3 kernel shape differs from dense_1"; return false; }
  w->wk2 = k.f; w->bk2 = b.f;
  if ((int64_t)w->bq2.size() != hq2 || (int64_t)w->bk1.size() != hq || (int64_t)w->bk2.size() != hq2) { *err = "nonlinear_attention: bias lengths do not match the kernels"; return false; }
  // PReLU: `_alpha * tf.minimum(0.0, x)` = the Mul ops `mul` (candidate side) and `mul_1` (sequence side)
  if (!variable(g, A + "/prelu_q", &t) && !operand_of(g, A + "/mul", -1, &t)) { *err = "no PReLU slope nonlinear_attention/prelu_q"; return false; }
  if (!take(t, hq, &w->aq, "nonlinear_attention/prelu_q", err)) return false;
  if (!variable(g, A + "/prelu_k", &t) && !operand_of(g, A + "/mul_1", -1, &t)) { *err = "no PReLU slope nonlinear_attention/prelu_k"; return false; }
  if (!take(t, hq, &w->ak, "nonlinear_attention/prelu_k", err)) return false;
  int64_t n_in = (int64_t)w->e + w->d;
  for (int l = 0; l < 4; ++l) {
    const std::string S = std::to_string(l + 1) + "_dnn";
    if (!dense_layer(g, S + "/fc", &k, l < 3 ? &b : nullptr, err)) return false;
    if (k.shape[0] != n_in) { *err = S + "/fc kernel has " + std::to_string(k.shape[0]) + " input rows, expected " + std::to_string(n_in); return false; }
    const int64_t n_out = k.shape[1];
    w->w[l] = k.f;
    if (l == 3) { if (n_out != 1) { *err = "4_dnn/fc must have one output"; return false; } break; }
    if (!take(b, n_out, &w->b[l], S + "/fc bias", err)) return false;
    // batch norm, inference form: y = x * scale + shift with scale = gamma * rsqrt(moving_variance + eps) the
    // second operand of batchnorm/mul_1 and shift = beta - moving_mean * scale the second operand of batchnorm/add_1
    Tensor sc, sh;
    if (!operand_of(g, S + "/bn/batchnorm/mul_1", 1, &sc)) {
      Tensor gamma, var, eps;
      if (!variable(g, S + "/bn/gamma", &gamma) || !variable(g, S + "/bn/moving_variance", &var)) { *err = "no batch-norm scale for " + S; return false; }
      float e = 1e-3f;  // tf.layers.batch_normalization default
      if (operand_of(g, S + "/bn/batchnorm/add", 1, &eps) && eps.f.size() == 1) e = eps.f[0];
      sc = gamma;
      if (var.f.size() != gamma.f.size()) { *err = S + "/bn: gamma and moving_variance differ in length"; return false; }
      for (size_t j = 0; j < sc.f.size(); ++j) sc.f[j] = gamma.f[j] * (1.0f / std::sqrt(var.f[j] + e));
    }
    if (!operand_of(g, S + "/bn/batchnorm/add_1", 1, &sh)) {
      Tensor beta, mean;
      if (!variable(g, S + "/bn/beta", &beta) || !variable(g, S + "/bn/moving_mean", &mean) || beta.f.size() != sc.f.size() ||
          mean.f.size() != sc.f.size()) { *err = "no batch-norm shift for " + S; return false; }
      sh = beta;
      for (size_t j = 0; j < sh.f.size(); ++j) sh.f[j] = beta.f[j] - mean.f[j] * sc.f[j];
    }
    if (!take(sc, n_out, &w->bn_scale[l], S + " batch-norm scale", err) || !take(sh, n_out, &w->bn_shift[l], S + " batch-norm shift", err)) return false;
    if (!variable(g, S + "/prelu", &t) && !operand_of(g, S + "/mul", -1, &t)) { *err = "no PReLU slope " + S + "/prelu"; return false; }
    if (!take(t, n_out, &w->alpha[l], S + "/prelu", err)) return false;
    n_in = n_out;
  }
  return true;
}

}  // namespace nann_gd
