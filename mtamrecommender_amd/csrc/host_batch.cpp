// libmtam_host.so: record parsing and batch packing on the host (see include/mtam_host.h).
// Plain C++17, no GPU code; built with g++ by the same Makefile as the HIP library.
#include "../../include/mtam_host.h"

#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

struct MtamRecordSet {
  std::vector<int64_t> offsets;   // n + 1
  std::vector<int32_t> user_id, target_id, target_category, length;
  std::vector<float> target_time;
  std::vector<int32_t> item, category, position;
  std::vector<float> time, timelast, timenow;
  int max_length = 0;
};

namespace {

void set_err(char *err, int err_len, const char *fmt, ...) {
  if (!err || err_len <= 0) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err, (size_t)err_len, fmt, ap);
  va_end(ap);
}

// A tiny recursive-descent reader for one line: "(" value {"," value} ")" with value = number | "[" numbers "]".
struct Cursor {
  const char *p, *end;
  void skip() {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
  }
  bool eat(char c) {
    skip();
    if (p < end && *p == c) {
      ++p;
      return true;
    }
    return false;
  }
  bool number(double &v) {
    skip();
    if (p >= end) return false;
    char *q = nullptr;
    errno = 0;
    v = strtod(p, &q);       // the line is NUL- or newline-terminated inside a NUL-terminated buffer
    if (q == p || q > end) return false;
    p = q;
    return true;
  }
  bool list(std::vector<double> &out) {
    out.clear();
    if (!eat('[')) return false;
    if (eat(']')) return true;
    for (;;) {
      double v;
      if (!number(v)) return false;
      out.push_back(v);
      if (eat(',')) {
        if (eat(']')) return true;      // trailing comma
        continue;
      }
      return eat(']');
    }
  }
};

bool as_i32(double v, int32_t &out) {
  if (!(v >= -2147483648.0 && v <= 2147483647.0) || v != (double)(long long)v) return false;
  out = (int32_t)v;
  return true;
}

bool parse_line(const char *b, const char *e, MtamRecordSet &rs, std::string &why) {
  Cursor c{b, e};
  std::vector<double> lists[6], target;
  double user, length;
  if (!c.eat('(')) return why = "expected '('", false;
  if (!c.number(user) || !c.eat(',')) return why = "user id", false;
  for (int i = 0; i < 6; ++i)
    if (!c.list(lists[i]) || !c.eat(',')) return why = "list " + std::to_string(i + 1), false;
  if (!c.list(target) || target.size() != 3 || !c.eat(',')) return why = "target triple", false;
  if (!c.number(length)) return why = "length", false;
  c.eat(',');
  if (!c.eat(')')) return why = "expected ')'", false;
  c.skip();
  if (c.p != c.end) return why = "trailing characters", false;
  const size_t n = lists[0].size();
  for (int i = 1; i < 6; ++i)
    if (lists[i].size() != n) return why = "the six lists differ in length", false;
  int32_t u, len, tid, tcat;
  if (!as_i32(user, u) || !as_i32(length, len) || !as_i32(target[0], tid) || !as_i32(target[1], tcat))
    return why = "non-integer id", false;
  // order in the tuple: items, categories, time, timelast, timenow, positions
  for (size_t j = 0; j < n; ++j) {
    int32_t it, ca, po;
    if (!as_i32(lists[0][j], it) || !as_i32(lists[1][j], ca) || !as_i32(lists[5][j], po))
      return why = "non-integer id in a list", false;
    rs.item.push_back(it);
    rs.category.push_back(ca);
    rs.position.push_back(po);
    rs.time.push_back((float)lists[2][j]);
    rs.timelast.push_back((float)lists[3][j]);
    rs.timenow.push_back((float)lists[4][j]);
  }
  rs.user_id.push_back(u);
  rs.target_id.push_back(tid);
  rs.target_category.push_back(tcat);
  rs.target_time.push_back((float)target[2]);
  rs.length.push_back(len);
  rs.offsets.push_back((int64_t)rs.item.size());
  if ((int)n > rs.max_length) rs.max_length = (int)n;
  return true;
}

MtamRecordSet *parse_buffer(const char *text, long len, char *err, int err_len) {
  MtamRecordSet *rs = new MtamRecordSet();
  rs->offsets.push_back(0);
  const char *p = text, *end = text + len;
  long line_no = 0;
  while (p < end) {
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    const char *e = nl ? nl : end;
    ++line_no;
    const char *q = p;
    while (q < e && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
    if (q < e) {
      std::string why;
      if (!parse_line(q, e, *rs, why)) {
        set_err(err, err_len, "line %ld: %s", line_no, why.c_str());
        delete rs;
        return nullptr;
      }
    }
    p = nl ? nl + 1 : end;
  }
  return rs;
}

uint64_t splitmix64(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

}  // namespace

extern "C" MtamRecordSet *mtam_records_parse_text(const char *text, long len, char *err, int err_len) {
  if (!text || len < 0) {
    set_err(err, err_len, "null text");
    return nullptr;
  }
  std::string copy(text, (size_t)len);       // strtod needs a terminated buffer
  return parse_buffer(copy.c_str(), len, err, err_len);
}

extern "C" MtamRecordSet *mtam_records_parse_file(const char *path, char *err, int err_len) {
  FILE *f = path ? fopen(path, "rb") : nullptr;
  if (!f) {
    set_err(err, err_len, "cannot open %s: %s", path ? path : "(null)", strerror(errno));
    return nullptr;
  }
  std::string buf;
  char chunk[1 << 16];
  size_t got;
  while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) buf.append(chunk, got);
  fclose(f);
  return parse_buffer(buf.c_str(), (long)buf.size(), err, err_len);
}

extern "C" MtamRecordSet *mtam_records_from_arrays(long n, const int64_t *offsets, const int32_t *user_id,
                                                   const int32_t *item, const int32_t *category,
                                                   const float *time, const float *timelast,
                                                   const float *timenow, const int32_t *position,
                                                   const int32_t *target_id, const int32_t *target_category,
                                                   const float *target_time, const int32_t *length, char *err,
                                                   int err_len) {
  if (n < 0 || !offsets || (n > 0 && (!user_id || !target_id || !target_category || !target_time || !length))) {
    set_err(err, err_len, "records_from_arrays: null argument");
    return nullptr;
  }
  if (offsets[0] != 0) {
    set_err(err, err_len, "records_from_arrays: offsets[0] must be 0");
    return nullptr;
  }
  for (long i = 0; i < n; ++i)
    if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0x7fffffff) {
      set_err(err, err_len, "records_from_arrays: offsets must be non-decreasing (record %ld)", i);
      return nullptr;
    }
  const int64_t total = offsets[n];
  if (total > 0 && (!item || !category || !time || !timelast || !timenow || !position)) {
    set_err(err, err_len, "records_from_arrays: null event array");
    return nullptr;
  }
  MtamRecordSet *rs = new MtamRecordSet();
  rs->offsets.assign(offsets, offsets + n + 1);
  rs->user_id.assign(user_id, user_id + n);
  rs->target_id.assign(target_id, target_id + n);
  rs->target_category.assign(target_category, target_category + n);
  rs->target_time.assign(target_time, target_time + n);
  rs->length.assign(length, length + n);
  rs->item.assign(item, item + total);
  rs->category.assign(category, category + total);
  rs->position.assign(position, position + total);
  rs->time.assign(time, time + total);
  rs->timelast.assign(timelast, timelast + total);
  rs->timenow.assign(timenow, timenow + total);
  for (long i = 0; i < n; ++i) {
    const int len = (int)(offsets[i + 1] - offsets[i]);
    if (len > rs->max_length) rs->max_length = len;
  }
  return rs;
}

extern "C" void mtam_records_free(MtamRecordSet *rs) { delete rs; }
extern "C" long mtam_records_count(const MtamRecordSet *rs) { return rs ? (long)rs->user_id.size() : 0; }
extern "C" int mtam_records_max_length(const MtamRecordSet *rs) { return rs ? rs->max_length : 0; }

extern "C" int mtam_records_get(const MtamRecordSet *rs, long i, int cap, int32_t *user_id, int32_t *item,
                                int32_t *category, float *time, float *timelast, float *timenow,
                                int32_t *position, int32_t *target_id, int32_t *target_category,
                                float *target_time, int32_t *length) {
  if (!rs || i < 0 || i >= (long)rs->user_id.size()) return -1;
  const int64_t o = rs->offsets[i];
  const int n = (int)(rs->offsets[i + 1] - o);
  if (n > cap) return -2;
  *user_id = rs->user_id[i];
  *target_id = rs->target_id[i];
  *target_category = rs->target_category[i];
  *target_time = rs->target_time[i];
  *length = rs->length[i];
  memcpy(item, rs->item.data() + o, sizeof(int32_t) * n);
  memcpy(category, rs->category.data() + o, sizeof(int32_t) * n);
  memcpy(position, rs->position.data() + o, sizeof(int32_t) * n);
  memcpy(time, rs->time.data() + o, sizeof(float) * n);
  memcpy(timelast, rs->timelast.data() + o, sizeof(float) * n);
  memcpy(timenow, rs->timenow.data() + o, sizeof(float) * n);
  return n;
}

extern "C" int mtam_pack_batch(const MtamRecordSet *rs, const int64_t *index, int B, int L,
                               const MtamArenaLayout *lay, const MtamTableRows *rows, float lr, int32_t *arena,
                               char *err, int err_len) {
  if (!rs || !index || !lay || !rows || !arena || B <= 0 || L <= 0) {
    set_err(err, err_len, "pack_batch: bad arguments");
    return -1;
  }
  const long n_rec = (long)rs->user_id.size();
  memset(arena, 0, sizeof(int32_t) * (size_t)lay->words);
  float *farena = reinterpret_cast<float *>(arena);
  for (int b = 0; b < B; ++b) {
    const int64_t i = index[b];
    if (i < 0 || i >= n_rec) {
      set_err(err, err_len, "pack_batch: record index %lld out of range", (long long)i);
      return -2;
    }
    const int64_t o = rs->offsets[i];
    const int n = (int)(rs->offsets[i + 1] - o);
    const int len = rs->length[i];
    if (n > L || len > L) {
      // np.pad with a negative width raises in the reference as well
      set_err(err, err_len, "record %lld: length %d exceeds length_of_user_history %d", (long long)i,
              n > len ? n : len, L);
      return -3;
    }
    if (len < 2) {
      set_err(err, err_len, "record %lld: seq_length %d must be in [2, %d]", (long long)i, len, L);
      return -3;
    }
    const int32_t u = rs->user_id[i], t = rs->target_id[i];
    if (u < 0 || u >= rows->user_rows) {
      set_err(err, err_len, "record %lld: user_id %d out of range [0, %d)", (long long)i, u, rows->user_rows);
      return -4;
    }
    if (t < 0 || t >= rows->item_rows) {
      set_err(err, err_len, "record %lld: target_item_id %d out of range [0, %d)", (long long)i, t, rows->item_rows);
      return -4;
    }
    const int32_t *it = rs->item.data() + o, *ca = rs->category.data() + o, *po = rs->position.data() + o;
    for (int j = 0; j < n; ++j) {
      if (it[j] < 0 || it[j] >= rows->item_rows || ca[j] < 0 || ca[j] >= rows->category_rows || po[j] < 0 ||
          po[j] >= rows->position_rows) {
        set_err(err, err_len, "record %lld, event %d: id out of range (item %d / %d, category %d / %d, position %d / %d)",
                (long long)i, j, it[j], rows->item_rows, ca[j], rows->category_rows, po[j], rows->position_rows);
        return -4;
      }
    }
    arena[lay->user_id + b] = u;
    arena[lay->target_item_id + b] = t;
    arena[lay->seq_length + b] = len;
    farena[lay->target_item_time + b] = rs->target_time[i];
    const size_t row = (size_t)b * (size_t)L;
    memcpy(arena + lay->item_list + row, it, sizeof(int32_t) * n);
    memcpy(arena + lay->category_list + row, ca, sizeof(int32_t) * n);
    memcpy(arena + lay->position_list + row, po, sizeof(int32_t) * n);
    memcpy(farena + lay->time_list + row, rs->time.data() + o, sizeof(float) * n);
    memcpy(farena + lay->timelast_list + row, rs->timelast.data() + o, sizeof(float) * n);
    memcpy(farena + lay->timenow_list + row, rs->timenow.data() + o, sizeof(float) * n);
  }
  farena[lay->lr] = lr;
  return 0;
}

extern "C" void mtam_shuffle_index(int64_t *index, long n, uint64_t seed) {
  if (!index) return;
  for (long i = 0; i < n; ++i) index[i] = i;
  uint64_t s = seed;
  for (long i = n - 1; i > 0; --i) {
    const uint64_t j = splitmix64(s) % (uint64_t)(i + 1);
    const int64_t t = index[i];
    index[i] = index[(long)j];
    index[(long)j] = t;
  }
}

// CRC-32C (Castagnoli), slicing-by-8: the checksum of TensorFlow's checkpoint bundle files (util/tf_bundle.py; a
// byte loop in Python takes minutes for a catalog-sized table)
extern "C" uint32_t mtam_crc32c(const void *data, size_t n, uint32_t crc) {
  static uint32_t table[8][256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
    ready = true;
  }
  const unsigned char *p = static_cast<const unsigned char *>(data);
  uint32_t c = crc ^ 0xFFFFFFFFu;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
        table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

extern "C" int mtam_host_version(void) { return 1; }
