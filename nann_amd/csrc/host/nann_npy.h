// nann_npy.h -- the .npy reader behind HugeConst (UO/huge_const_op/huge_const_op.cc:85-182, npy.h:541-571) and the
// scorer-model loader, as a pure host header: it decodes a file image IN MEMORY, so the same code runs in libnann_hip.so
// (nann_huge_const_load) and, under AddressSanitizer / UBSan, in the parser fuzzers of the CPU suite (tests/fuzz/).
//
// Files come from outside (the reference's build_hnsw_index.py / np.save; a serving host's model directory), so every
// length is checked before it is used: header length against the image, dict keys present, shape rank and the PRODUCT of
// its dims against int64 and against the bytes that are actually there -- a header that promises more than the file holds
// is "truncated npy payload" before anything is allocated.  Reference behaviour kept: formats 1.0 and 2.0 (3.0 has the
// same layout as 2.0 with a utf-8 header), Fortran order refused (huge_const_op.cc:108-109), descr compared with the
// requested dtype (:117-147), each dim with the `shape` attr (:111-115).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace nann_npy {

// status codes = include/nann_hip.h's (kept numerically equal: the caller passes them through)
enum { kOk = 0, kUnsupported = 102, kIo = 104, kDtypeMismatch = 105, kShapeMismatch = 106 };
// dtype codes = enum nann_dtype
enum { kF16 = 0, kBF16 = 1, kF32 = 2, kI32 = 3, kI64 = 4, kF64 = 5 };

inline const char* descr_of(int dtype) {
  switch (dtype) {
    case kF16: return "<f2";
    case kF32: return "<f4";
    case kF64: return "<f8";
    case kI32: return "<i4";
    case kI64: return "<i8";
    default: return nullptr;  // (bf16 has no numpy descr)
  }
}
inline int elem_bytes(int dtype) {
  static const int esz[6] = {2, 2, 4, 4, 8, 8};
  return dtype >= 0 && dtype < 6 ? esz[dtype] : 0;
}

struct Header {
  int major = 0;
  std::string descr;
  bool fortran = false;
  std::vector<int64_t> shape;
  size_t data_offset = 0;  // where the payload starts in the image
  int64_t count = 0;       // product of the dims (1 for a 0-d array)
};

constexpr size_t kMaxHeaderBytes = 1u << 20;  // numpy writes ~100 bytes; format 2.0 allows 4 GiB
constexpr int kMaxRank = 32;                  // NPY_MAXDIMS

// value of `'key':` in the header dict: position of the first non-blank character behind the colon, or npos
inline size_t dict_value(const std::string& hdr, const char* key) {
  const size_t p = hdr.find(key);
  if (p == std::string::npos) return p;
  const size_t c = hdr.find(':', p + std::strlen(key));
  if (c == std::string::npos) return c;
  return hdr.find_first_not_of(" \t", c + 1);
}

inline int parse_header(const unsigned char* data, size_t n, Header* h, std::string* err) {
  auto fail = [&](int code, const std::string& m) { *err = m; return code; };
  if (n < 10 || std::memcmp(data, "\x93NUMPY", 6) != 0) return fail(kIo, "not an npy file");
  h->major = data[6];
  size_t hlen = 0, at = 0;
  if (h->major == 1) {
    hlen = (size_t)data[8] | ((size_t)data[9] << 8);
    at = 10;
  } else if (h->major == 2 || h->major == 3) {  // npy.h:541-571 accepts 1.0 and 2.0; 3.0 = 2.0's layout
    if (n < 12) return fail(kIo, "truncated npy header");
    hlen = (size_t)data[8] | ((size_t)data[9] << 8) | ((size_t)data[10] << 16) | ((size_t)data[11] << 24);
    at = 12;
  } else {
    return fail(kIo, "unsupported npy version");
  }
  if (hlen > kMaxHeaderBytes) return fail(kIo, "npy header longer than 1 MiB");
  if (hlen > n - at) return fail(kIo, "truncated npy header");
  const std::string hdr(reinterpret_cast<const char*>(data + at), hlen);
  h->data_offset = at + hlen;
  // 'descr': '<f2'
  size_t p = dict_value(hdr, "'descr'");
  if (p == std::string::npos) return fail(kIo, "npy header without descr");
  if (hdr[p] != '\'' && hdr[p] != '"') return fail(kUnsupported, "npy descr is not a plain dtype string (structured arrays are not supported)");
  const size_t q1 = hdr.find(hdr[p], p + 1);
  if (q1 == std::string::npos) return fail(kIo, "unterminated npy descr");
  h->descr = hdr.substr(p + 1, q1 - p - 1);
  if (!h->descr.empty() && (h->descr[0] == '|' || h->descr[0] == '=')) h->descr[0] = '<';  // byte order not applicable / native
  // 'fortran_order': False
  p = dict_value(hdr, "'fortran_order'");
  if (p == std::string::npos) return fail(kIo, "npy header without fortran_order");
  if (hdr.compare(p, 4, "True") == 0) h->fortran = true;
  else if (hdr.compare(p, 5, "False") == 0) h->fortran = false;
  else return fail(kIo, "npy fortran_order is neither True nor False");
  // 'shape': (3, 4)
  p = dict_value(hdr, "'shape'");
  if (p == std::string::npos) return fail(kIo, "npy header without shape");
  if (hdr[p] != '(') return fail(kIo, "bad npy shape");
  const size_t s1 = hdr.find(')', p);
  if (s1 == std::string::npos) return fail(kIo, "bad npy shape");
  h->shape.clear();
  h->count = 1;
  for (size_t i = p + 1; i < s1;) {
    while (i < s1 && (hdr[i] == ' ' || hdr[i] == ',')) ++i;
    if (i >= s1) break;
    int64_t v = 0;
    size_t j = i;
    while (j < s1 && hdr[j] >= '0' && hdr[j] <= '9') {
      if (v > (INT64_MAX - 9) / 10) return fail(kIo, "npy shape overflows int64");
      v = v * 10 + (hdr[j] - '0');
      ++j;
    }
    if (j < s1 && hdr[j] == 'L') ++j;  // python 2 longs: (3L, 4L)
    if (j == i) return fail(kIo, "bad npy shape");
    if ((int)h->shape.size() >= kMaxRank) return fail(kIo, "npy shape has more than 32 dims");
    if (v != 0 && h->count > INT64_MAX / v) return fail(kIo, "npy shape overflows int64");
    h->count *= v;
    h->shape.push_back(v);
    i = j;
  }
  return kOk;
}

// np.ndarray.astype for the casts the reference's Python wrapper performs (model_util.py:116-119)
inline uint16_t f32_to_f16_rne(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
  uint32_t man = x & 0x7fffffu;
  if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    man |= 0x800000u;
    const int shift = 14 - exp;
    uint32_t hm = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) ++hm;
    return (uint16_t)(sign | hm);
  }
  uint32_t out = sign | ((uint32_t)exp << 10) | (man >> 13);
  const uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (out & 1))) ++out;
  return (uint16_t)out;
}

template <typename T> inline T load_as(const unsigned char* p) { T v; std::memcpy(&v, p, sizeof(T)); return v; }

// payload (`count` elements of `from`) -> `to`; false: no such cast is offered
inline bool cast_payload(const std::string& from, int to, const unsigned char* in, int64_t count, std::vector<char>* out) {
  out->resize((size_t)(count * elem_bytes(to) > 0 ? count * elem_bytes(to) : 1));
  if (from == "<i8" && to == kI32) {
    int32_t* b = reinterpret_cast<int32_t*>(out->data());
    for (int64_t i = 0; i < count; ++i) b[i] = (int32_t)(uint32_t)(uint64_t)load_as<int64_t>(in + 8 * i);  // astype wraps
    return true;
  }
  if (from == "<i4" && to == kI64) {
    int64_t* b = reinterpret_cast<int64_t*>(out->data());
    for (int64_t i = 0; i < count; ++i) b[i] = load_as<int32_t>(in + 4 * i);
    return true;
  }
  if (from == "<f4" && to == kF16) {
    uint16_t* b = reinterpret_cast<uint16_t*>(out->data());
    for (int64_t i = 0; i < count; ++i) b[i] = f32_to_f16_rne(load_as<float>(in + 4 * i));
    return true;
  }
  if (from == "<f8" && to == kF32) {
    float* b = reinterpret_cast<float*>(out->data());
    for (int64_t i = 0; i < count; ++i) b[i] = (float)load_as<double>(in + 8 * i);
    return true;
  }
  return false;  // (f64 -> f16 through f32 would double-round: not offered)
}

// A whole file image -> its payload in `expect_dtype`.  Without a cast the payload is NOT copied: *payload / *payload_bytes
// point into `data`; with one they point into *converted.  expect_shape may be null (any shape).
inline int decode(const unsigned char* data, size_t n, int expect_dtype, const int64_t* expect_shape, int expect_rank,
                  bool allow_cast, const unsigned char** payload, size_t* payload_bytes, std::vector<char>* converted,
                  std::vector<int64_t>* out_shape, std::string* err) {
  auto fail = [&](int code, const std::string& m) { *err = m; return code; };
  Header h;
  int rc = parse_header(data, n, &h, err);
  if (rc) return rc;
  if (h.fortran) return fail(kUnsupported, "Fortran order NOT supported.");  // huge_const_op.cc:108-109
  const char* want = descr_of(expect_dtype);
  if (!want) return fail(kUnsupported, "Unsupported DataType.");             // :143-146
  const bool need_cast = h.descr != want;
  if (need_cast && !allow_cast) return fail(kDtypeMismatch, "DataType mismatch: " + h.descr + "!=" + want);  // :117-121
  if (expect_shape) {
    if ((int)h.shape.size() != expect_rank) return fail(kShapeMismatch, "rank mismatch");
    for (int i = 0; i < expect_rank; ++i)
      if (h.shape[(size_t)i] != expect_shape[i])
        return fail(kShapeMismatch, "attr_shape and np_shape NOT match in dim " + std::to_string(i));  // :111-115
  }
  int file_esz = elem_bytes(expect_dtype);
  if (need_cast) {
    if (h.descr != "<i8" && h.descr != "<i4" && h.descr != "<f4" && h.descr != "<f8" && h.descr != "<f2")
      return fail(kDtypeMismatch, "DataType mismatch: " + h.descr + "!=" + want);
    file_esz = h.descr[2] - '0';
  }
  const size_t avail = n - h.data_offset;
  if (h.count < 0 || (uint64_t)h.count > (uint64_t)avail / (uint64_t)file_esz)
    return fail(kIo, "truncated npy payload");  // (also every count whose byte size would overflow)
  const unsigned char* src = data + h.data_offset;
  if (need_cast) {
    if (!cast_payload(h.descr, expect_dtype, src, h.count, converted))
      return fail(kDtypeMismatch, std::string("no cast from ") + h.descr + " to " + want);
    converted->resize((size_t)(h.count * elem_bytes(expect_dtype)));
    *payload = reinterpret_cast<const unsigned char*>(converted->data());
    *payload_bytes = converted->size();
  } else {
    *payload = src;
    *payload_bytes = (size_t)h.count * (size_t)file_esz;
  }
  if (out_shape) *out_shape = h.shape;
  return kOk;
}

}  // namespace nann_npy
