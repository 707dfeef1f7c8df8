// nann_graphdef_c.cpp -- CPU-side hook onto the frozen-GraphDef reader (nann_graphdef.h) for the tests that run
// without a GPU: the same parser and weight extraction nann_model_load uses, behind a plain C function in
// libnann_host.so.
#include <cstdio>
#include <fstream>
#include <iterator>

#include "nann_graphdef_text.h"
#include "nann_blaze_options.h"
#include "nann_npy.h"

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

// ---- entry points of the parser fuzzers (tests/fuzz/fuzz_parsers.py; libnann_host_asan.so = this file under
// -fsanitize=address,undefined).  Each runs ONE parser of external bytes on a memory image and everything built on its result,
// and returns 0 (accepted) / 1 (rejected with a message); a crash, an out-of-bounds read or an overflow is the sanitizer's to report.

// format: 0 = as BlazeXlaOp reads a graph_def (text first, then binary), 1 = binary GraphDef, 2 = binary SavedModel, 3 = text
int nann_fuzz_graphdef(const uint8_t* data, int64_t n, int32_t format) {
  nann_gd::Graph g;
  std::string msg;
  bool ok = false;
  if (format == 0) ok = nann_gd::parse_graph_any(data, (size_t)n, &g, &msg);
  else if (format == 1) ok = nann_gd::parse_graph(data, (size_t)n, &g, &msg);
  else if (format == 2) ok = nann_gd::parse_saved_model(data, (size_t)n, &g, &msg);
  else ok = nann_gd::parse_graph_text(reinterpret_cast<const char*>(data), (size_t)n, &g, &msg);
  if (!ok) return 1;
  nann_gd::AttnWeights w;
  (void)nann_gd::extract_attention(g, &w, &msg);  // the weight extraction walks whatever graph was accepted
  const std::string js = nann_gd::graph_to_json(g);
  return js.empty() ? 1 : 0;
}

// the .npy decoder on a file image; *sum = a checksum over the decoded payload (every byte of it is read)
int nann_fuzz_npy(const uint8_t* data, int64_t n, int32_t expect_dtype, int32_t allow_cast, const int64_t* expect_shape,
                  int32_t expect_rank, uint64_t* sum, char* err, int32_t err_len) {
  const unsigned char* payload = nullptr;
  size_t bytes = 0;
  std::vector<char> conv;
  std::vector<int64_t> shape;
  std::string msg;
  const int rc = nann_npy::decode(data, (size_t)n, expect_dtype, expect_shape, expect_rank, allow_cast != 0, &payload, &bytes, &conv, &shape, &msg);
  if (rc) {
    if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "%s", msg.c_str());
    return rc;
  }
  uint64_t s = 1469598103934665603ull;
  for (size_t i = 0; i < bytes; ++i) s = (s ^ payload[i]) * 1099511628211ull;
  for (int64_t d : shape) s = (s ^ (uint64_t)d) * 1099511628211ull;
  if (sum) *sum = s;
  return 0;
}

int nann_fuzz_blaze_options(const char* data, int64_t n) {
  nann_gd::BlazeOptions o;
  std::string msg;
  return nann_gd::parse_blaze_options_text(data, (size_t)n, &o, &msg) ? 0 : 1;
}

// what HugeConst's loader makes of a FILE (the CPU tests of npy 2.0 / Fortran order / truncation: no GPU needed).
// shape[32], *rank, *payload_bytes; returns the nann_status nann_huge_const_load would return, message in err.
int nann_host_npy_info(const char* path, int32_t expect_dtype, int32_t allow_cast, int64_t* shape, int32_t* rank,
                       int64_t* payload_bytes, char* err, int32_t err_len) {
  std::ifstream f(path, std::ifstream::binary);
  if (!f) { if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "Fail to open file: %s", path); return nann_npy::kIo; }
  const std::string image((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const unsigned char* payload = nullptr;
  size_t bytes = 0;
  std::vector<char> conv;
  std::vector<int64_t> sh;
  std::string msg;
  const int rc = nann_npy::decode(reinterpret_cast<const unsigned char*>(image.data()), image.size(), expect_dtype, nullptr, 0, allow_cast != 0,
                                  &payload, &bytes, &conv, &sh, &msg);
  if (rc) { if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "%s", msg.c_str()); return rc; }
  if (rank) *rank = (int32_t)sh.size();
  for (size_t i = 0; i < sh.size() && i < 32 && shape; ++i) shape[i] = sh[i];
  if (payload_bytes) *payload_bytes = (int64_t)bytes;
  return 0;
}

}  // extern "C"
