// nann_graphdef_c.cpp -- CPU-side hook onto the frozen-GraphDef reader (nann_graphdef.h) for the tests that run
// without a GPU: the same parser and weight extraction nann_model_load uses, behind a plain C function in
// libnann_host.so.
#include <cstdio>
#include <fstream>
#include <iterator>

#include "nann_graphdef_text.h"
#include "nann_blaze_options.h"

extern "C" {

// The 26 tensors of the reference's scorer model in nann_attn_desc order: wq1 bq1 aq wq2 bq2 wk1 bk1 ak wk2 bk2,
// then per DNN layer l = 0..2: w[l] b[l] bn_scale[l] bn_shift[l] alpha[l], then w[3].
// counts[26] always; flat (may be NULL) receives the tensors back to back.  Returns 0, or 1 with a message in err.
int nann_graphdef_attention(const char* path, int64_t counts[26], float* flat, int32_t* d, int32_t* e, char* err,
                            int32_t err_len) {
  auto fail = [&](const std::string& m) { if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "%s", m.c_str()); return 1; };
  std::ifstream f(path, std::ifstream::binary);
  if (!f) return fail(std::string("Fail to open file: ") + path);
  const std::string bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  nann_gd::Graph g;
  std::string msg;
  // text first, then binary: the reference's order (blaze_xla_kernel.cc:169-175)
  if (!nann_gd::parse_graph_any(reinterpret_cast<const uint8_t*>(bytes.data()), bytes.size(), &g, &msg)) return fail(msg);
  nann_gd::AttnWeights w;
  if (!nann_gd::extract_attention(g, &w, &msg)) return fail(msg);
  const std::vector<float>* v[26] = {&w.wq1, &w.bq1, &w.aq, &w.wq2, &w.bq2, &w.wk1, &w.bk1, &w.ak, &w.wk2, &w.bk2,
                                     &w.w[0], &w.b[0], &w.bn_scale[0], &w.bn_shift[0], &w.alpha[0],
                                     &w.w[1], &w.b[1], &w.bn_scale[1], &w.bn_shift[1], &w.alpha[1],
                                     &w.w[2], &w.b[2], &w.bn_scale[2], &w.bn_shift[2], &w.alpha[2], &w.w[3]};
  for (int i = 0; i < 26; ++i) {
    counts[i] = (int64_t)v[i]->size();
    if (flat) { std::memcpy(flat, v[i]->data(), v[i]->size() * 4); flat += v[i]->size(); }
  }
  if (d) *d = w.d;
  if (e) *e = w.e;
  return 0;
}

// The decoded graph of `path` as canonical JSON (nodes in file order: name, op, device, inputs, every attr by kind).
// format: 0 = as BlazeXlaOp reads a graph_def (text first, then binary GraphDef), 1 = binary GraphDef, 2 = binary
// SavedModel (first meta graph), 3 = text (GraphDef or SavedModel).  *need = bytes of the JSON incl. the NUL; written
// to out when it fits cap.  Returns 0, or 1 with a message in err.
int nann_graphdef_dump(const char* path, int32_t format, char* out, int64_t cap, int64_t* need, char* err, int32_t err_len) {
  auto fail = [&](const std::string& m) { if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "%s", m.c_str()); return 1; };
  std::ifstream f(path, std::ifstream::binary);
  if (!f) return fail(std::string("Fail to open file: ") + path);
  const std::string bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const uint8_t* d = reinterpret_cast<const uint8_t*>(bytes.data());
  nann_gd::Graph g;
  std::string msg;
  bool ok = false;
  if (format == 0) ok = nann_gd::parse_graph_any(d, bytes.size(), &g, &msg);
  else if (format == 1) ok = nann_gd::parse_graph(d, bytes.size(), &g, &msg);
  else if (format == 2) ok = nann_gd::parse_saved_model(d, bytes.size(), &g, &msg);
  else if (format == 3) ok = nann_gd::parse_graph_text(bytes.data(), bytes.size(), &g, &msg);
  else return fail("unknown format");
  if (!ok) return fail(msg);
  const std::string js = nann_gd::graph_to_json(g);
  if (need) *need = (int64_t)js.size() + 1;
  if (out && cap >= (int64_t)js.size() + 1) std::memcpy(out, js.c_str(), js.size() + 1);
  return 0;
}

// BlazeXlaOp's blaze_option_path attr (nann_blaze_options.h), as nann_blaze_options_parse of libnann_hip.so reads it.
// out[8] = {wait_ms, run_mode, xla_compilation, auto_mixed_precision, disable_output_padding, n_warmup_batchsize,
// max_warmup_batchsize, from_file}.  Returns 0, or 1 with a message in err.
int nann_host_blaze_options(const char* attr, int32_t out[8], char* err, int32_t err_len) {
  nann_gd::BlazeOptions o;
  std::string msg;
  if (!nann_gd::parse_blaze_options_attr(attr ? attr : "", &o, &msg)) {
    if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "%s", msg.c_str());
    return 1;
  }
  const int32_t v[8] = {o.wait_ms, o.run_mode, o.xla_compilation, o.auto_mixed_precision, o.disable_output_padding,
                        o.n_warmup_batchsize, o.max_warmup_batchsize, o.from_file};
  std::memcpy(out, v, sizeof(v));
  return 0;
}

}  // extern "C"
