// nann_blaze_options.h -- the `blaze_option_path` attr of a BlazeXlaOp node.
//
// The reference reads it into a BlazeKernelOptions message (tensorflow/core/protobuf/config.proto:805-841) in
// BlazeXlaOp::ParseAttr (UO/blaze_op/blaze_xla_kernel.cc:156-167): the attr is first tried as the PATH of a text-format
// file (NANN_impls/nann/delivery/opt_default.conf is the one build_opt_graph.py passes), and when that does not read,
// the attr STRING ITSELF is parsed as text format; failing both is Internal("parse proto from ... failed").
// Of the message, the MI355X op keeps what still has a meaning without XLA and a nested session: `wait_ms` (the
// admission deadline of blaze_xla_kernel.cc:221-258) and `run_mode` (SKIP: :202-205, 183-188); the warm-up sizes,
// XLA / grappler switches and the nested ConfigProto are accepted and have no effect (there is no compilation step to
// warm up and rows are scored as they come).  Text format is read by the generic reader the frozen-GraphDef loader
// uses (nann_graphdef_text.h); a top-level field BlazeKernelOptions does not have fails the parse, as protobuf's
// TextFormat does.
#pragma once
#include <cerrno>
#include <fstream>
#include <iterator>

#include "nann_graphdef_text.h"

namespace nann_gd {

struct BlazeOptions {
  int32_t wait_ms = 0;
  int32_t run_mode = 0;  // 0 DEFAULT, 1 BENCHMARK, 2 SKIP (config.proto:833-837)
  int32_t xla_compilation = 0;
  int32_t auto_mixed_precision = 0;
  int32_t disable_output_padding = 0;
  int32_t n_warmup_batchsize = 0;
  int32_t max_warmup_batchsize = 0;
  int32_t from_file = 0;  // 1: the attr named a file that parsed; 0: the attr string itself was the message
};

inline bool blaze_int32(const TField& f, int32_t* out, std::string* err) {
  if (f.is_msg || f.quoted) { *err = "field '" + f.name + "': a number was expected"; return false; }
  errno = 0;
  char* end = nullptr;
  const long long v = std::strtoll(f.scalar.c_str(), &end, 0);
  if (end == f.scalar.c_str() || *end != 0 || errno == ERANGE || v < INT32_MIN || v > INT32_MAX) {
    *err = "field '" + f.name + "': '" + f.scalar + "' is not an int32";
    return false;
  }
  *out = (int32_t)v;
  return true;
}
inline bool blaze_bool(const TField& f, int32_t* out, std::string* err) {
  if (f.is_msg || f.quoted) { *err = "field '" + f.name + "': a bool was expected"; return false; }
  const std::string& s = f.scalar;
  if (s == "true" || s == "True" || s == "t" || s == "1") { *out = 1; return true; }
  if (s == "false" || s == "False" || s == "f" || s == "0") { *out = 0; return true; }
  *err = "field '" + f.name + "': '" + s + "' is not a bool";
  return false;
}

// one text-format BlazeKernelOptions message
inline bool parse_blaze_options_text(const char* data, size_t n, BlazeOptions* o, std::string* err) {
  TMsg root;
  if (!parse_text(data, n, &root, err)) return false;
  *o = BlazeOptions();
  // every field of the message (config.proto:806-840); `kind`: i = int32, b = bool, - = accepted, no effect here
  static const struct { const char* name; char kind; } kFields[] = {
      {"warmup_batchsize", 'w'}, {"xla_compilation", 'b'}, {"gemm_optimization", '-'}, {"auto_mixed_precision", 'b'},
      {"virtual_gpus_per_device", '-'}, {"inter_op_threads", '-'}, {"intra_op_threads", '-'}, {"no_warmup_inputs", '-'},
      {"use_single_threaded_executor", '-'}, {"callable_adapt_device", '-'}, {"gemm_dynamic_shape", '-'},
      {"max_batch_for_cpu", '-'}, {"cpu_buckets_nums", '-'}, {"disable_output_padding", 'b'}, {"unit_flops", '-'},
      {"debug_batchsize", '-'}, {"dump_timeline_interval", '-'}, {"config_proto", '-'}, {"warmup_value", '-'},
      {"run_mode", 'm'}, {"wait_ms", 'i'}};
  for (const TField& f : root.fields) {
    char kind = 0;
    for (const auto& k : kFields)
      if (f.name == k.name) { kind = k.kind; break; }
    if (!kind) { *err = "BlazeKernelOptions has no field named '" + f.name + "'"; return false; }
    if (kind == 'i') { if (!blaze_int32(f, &o->wait_ms, err)) return false; }
    else if (kind == 'w') {
      int32_t v = 0;
      if (!blaze_int32(f, &v, err)) return false;
      ++o->n_warmup_batchsize;
      if (v > o->max_warmup_batchsize) o->max_warmup_batchsize = v;
    } else if (kind == 'b') {
      int32_t* dst = f.name == "xla_compilation" ? &o->xla_compilation
                     : f.name == "auto_mixed_precision" ? &o->auto_mixed_precision : &o->disable_output_padding;
      if (!blaze_bool(f, dst, err)) return false;
    } else if (kind == 'm') {
      if (f.is_msg || f.quoted) { *err = "field 'run_mode': an enum value was expected"; return false; }
      if (f.scalar == "DEFAULT" || f.scalar == "0") o->run_mode = 0;
      else if (f.scalar == "BENCHMARK" || f.scalar == "1") o->run_mode = 1;
      else if (f.scalar == "SKIP" || f.scalar == "2") o->run_mode = 2;
      else { *err = "unknown RunMode '" + f.scalar + "'"; return false; }
    }
  }
  return true;
}

// the attr as BlazeXlaOp::ParseAttr reads it: a file first, then the string itself
inline bool parse_blaze_options_attr(const std::string& attr, BlazeOptions* o, std::string* err) {
  if (!attr.empty()) {
    std::ifstream f(attr, std::ifstream::binary);
    if (f) {
      const std::string bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      std::string ferr;
      if (f.good() || f.eof()) {
        if (parse_blaze_options_text(bytes.data(), bytes.size(), o, &ferr)) { o->from_file = 1; return true; }
      }
    }
  }
  std::string terr;
  if (parse_blaze_options_text(attr.data(), attr.size(), o, &terr)) { o->from_file = 0; return true; }
  *err = "parse proto from " + attr + " failed (" + terr + ")";
  return false;
}

}  // namespace nann_gd
