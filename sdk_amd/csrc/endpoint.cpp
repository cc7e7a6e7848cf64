// The request layer either side of the answer path: what lib/server's binary does around
// process_query, without the HTTP transport.
//   ServerState            lib/server/src/bin/server.rs:21-28   params + database + RwLock<HashMap<uuid, PublicParameters>>
//   POST /setup            bin/server.rs:71-94    JSON string of base64(public parameters) -> {"uuid":"..."}
//   POST /private-read     bin/server.rs:98-164   JSON list of base64 requests; request = uuid (36 bytes) || query when the
//                                                  params expand queries, public parameters || query otherwise; the
//                                                  reference answers them one by one (:152-158) -- here the list goes
//                                                  through sp_process_query_batch: <= 8 queries per database pass.
// Uses only the public C ABI (include/spiral_hip.h); framing (JSON list of strings, standard base64 with padding,
// what serde_json / the base64 crate emit) is plain host code.
#include <array>
#include <cerrno>
#include <cstring>
#include <sys/random.h>
#include <memory>
#include <mutex>
#include <random>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/spiral_hip.h"

extern "C" void sp_set_last_error_(const char* msg);  // capi.cpp

namespace {

constexpr size_t UUID_V4_STR_BYTES = 36;  // bin/server.rs:96

struct Fail {
  int rc;
  std::string msg;
};

const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";

std::string b64_encode(const uint8_t* d, size_t n) {
  std::string out;
  out.reserve((n + 2) / 3 * 4);
  for (size_t i = 0; i < n; i += 3) {
    const uint32_t a = d[i], b = i + 1 < n ? d[i + 1] : 0, c = i + 2 < n ? d[i + 2] : 0;
    const uint32_t v = (a << 16) | (b << 8) | c;
    out += B64[(v >> 18) & 63];
    out += B64[(v >> 12) & 63];
    out += i + 1 < n ? B64[(v >> 6) & 63] : '=';
    out += i + 2 < n ? B64[v & 63] : '=';
  }
  return out;
}

bool b64_decode(const char* s, size_t n, std::vector<uint8_t>& out) {
  // function-local static initialised by a lambda: thread-safe (sp_server_* may be entered concurrently)
  static const std::array<int8_t, 256> tab = [] {
    std::array<int8_t, 256> t;
    t.fill(-1);
    for (int i = 0; i < 64; i++) t[(unsigned char)B64[i]] = (int8_t)i;
    return t;
  }();
  if (n % 4 != 0) return false;
  out.clear();
  out.reserve(n / 4 * 3);
  for (size_t i = 0; i < n; i += 4) {
    int v[4];
    int pad = 0;
    for (int k = 0; k < 4; k++) {
      const char c = s[i + k];
      if (c == '=') {
        if (i + 4 != n || k < 2) return false;
        v[k] = 0;
        pad++;
      } else {
        if (pad) return false;
        v[k] = tab[(unsigned char)c];
        if (v[k] < 0) return false;
      }
    }
    const uint32_t w = ((uint32_t)v[0] << 18) | ((uint32_t)v[1] << 12) | ((uint32_t)v[2] << 6) | (uint32_t)v[3];
    out.push_back((uint8_t)(w >> 16));
    if (pad < 2) out.push_back((uint8_t)(w >> 8));
    if (pad < 1) out.push_back((uint8_t)w);
  }
  return true;
}

void skip_ws(const char*& p, const char* e) {
  while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
}
// a JSON string holding base64 text: no escapes other than \/ can occur in what serde_json / JSON.stringify emit for it
bool parse_string(const char*& p, const char* e, std::string& out) {
  skip_ws(p, e);
  if (p >= e || *p != '"') return false;
  p++;
  out.clear();
  while (p < e && *p != '"') {
    if (*p == '\\') {
      if (p + 1 < e && p[1] == '/') {
        out += '/';
        p += 2;
        continue;
      }
      return false;
    }
    out += *p++;
  }
  if (p >= e) return false;
  p++;
  return true;
}
bool parse_string_list(const char* s, size_t n, std::vector<std::string>& out) {
  const char* p = s;
  const char* e = s + n;
  skip_ws(p, e);
  if (p >= e || *p != '[') return false;
  p++;
  skip_ws(p, e);
  out.clear();
  if (p < e && *p == ']') {
    p++;
  } else {
    for (;;) {
      std::string item;
      if (!parse_string(p, e, item)) return false;
      out.push_back(std::move(item));
      skip_ws(p, e);
      if (p < e && *p == ',') {
        p++;
        continue;
      }
      if (p < e && *p == ']') {
        p++;
        break;
      }
      return false;
    }
  }
  skip_ws(p, e);
  return p == e;
}

std::string uuid_v4() {
  // The uuid is the only credential naming a client's resident public parameters in /private-read: 16 bytes from the
  // kernel's CSPRNG, as Uuid::new_v4 in the reference (lib/server/src/bin/server.rs:87), not a seeded Mersenne Twister.
  uint8_t b[16];
  size_t got = 0;
  while (got < sizeof(b)) {
    const ssize_t r = getrandom(b + got, sizeof(b) - got, 0);
    if (r < 0) {
      if (errno == EINTR) continue;
      throw Fail{SP_E_ARG, "getrandom failed"};
    }
    got += (size_t)r;
  }
  b[6] = (uint8_t)((b[6] & 0x0F) | 0x40);  // version 4
  b[8] = (uint8_t)((b[8] & 0x3F) | 0x80);  // RFC 4122 variant
  static const char hex[] = "0123456789abcdef";
  std::string s;
  for (int i = 0; i < 16; i++) {
    if (i == 4 || i == 6 || i == 8 || i == 10) s += '-';
    s += hex[b[i] >> 4];
    s += hex[b[i] & 15];
  }
  return s;
}

}  // namespace

struct sp_server {
  const sp_params_t* params = nullptr;
  const sp_db_t* db = nullptr;
  size_t setup_bytes = 0, query_bytes = 0, response_bytes = 0;
  bool expand_queries = true;
  mutable std::shared_mutex mu;  // bin/server.rs:26 RwLock<HashMap<String, PublicParameters>>
  std::unordered_map<std::string, std::shared_ptr<sp_pp_t>> pub_params;
};

template <typename F>
static int guarded_ep(F&& f) {
  try {
    f();
    return SP_OK;
  } catch (const Fail& e) {
    sp_set_last_error_(e.msg.c_str());
    return e.rc;
  } catch (const std::bad_alloc&) {
    sp_set_last_error_("host allocation failed");
    return SP_E_OOM;
  } catch (const std::exception& e) {
    sp_set_last_error_(e.what());
    return SP_E_ARG;
  }
}

static std::shared_ptr<sp_pp_t> deserialize_pp(const sp_server& S, const uint8_t* data, size_t len) {
  if (len != S.setup_bytes)  // bin/server.rs:83 assert_eq!(client_pub_params.len(), data.params.setup_bytes())
    throw Fail{SP_E_ARG, "public parameters: " + std::to_string(len) + " bytes, setup_bytes is " + std::to_string(S.setup_bytes)};
  sp_pp_t* pp = sp_pp_deserialize(S.params, data, len);
  if (!pp) throw Fail{SP_E_ARG, std::string("sp_pp_deserialize: ") + sp_last_error()};
  return std::shared_ptr<sp_pp_t>(pp, [](sp_pp_t* x) { sp_pp_free(x); });
}

// the body of private_read for a list of already base64-decoded requests
static void private_read(sp_server& S, const uint8_t* const* reqs, const size_t* lens, int n, uint8_t* out, size_t stride,
                         size_t* out_lens) {
  if (n == 0) return;
  if (stride < S.response_bytes) throw Fail{SP_E_ARG, "out_stride smaller than response_bytes"};
  std::vector<std::shared_ptr<sp_pp_t>> keep((size_t)n);
  std::vector<const sp_pp_t*> pps((size_t)n);
  std::vector<const uint8_t*> qs((size_t)n);
  std::vector<size_t> qlens((size_t)n, S.query_bytes);
  for (int i = 0; i < n; i++) {
    if (S.expand_queries) {
      // bin/server.rs:107-121: uuid || query, the uuid names public parameters uploaded through /setup
      if (lens[i] != UUID_V4_STR_BYTES + S.query_bytes)
        throw Fail{SP_E_ARG, "request " + std::to_string(i) + ": " + std::to_string(lens[i]) + " bytes, expected uuid (36) + query_bytes (" +
                                 std::to_string(S.query_bytes) + ")"};
      const std::string uuid((const char*)reqs[i], UUID_V4_STR_BYTES);
      std::shared_lock<std::shared_mutex> lk(S.mu);
      auto it = S.pub_params.find(uuid);
      if (it == S.pub_params.end()) throw Fail{SP_E_NOTFOUND, "request " + std::to_string(i) + ": unknown uuid " + uuid};  // Error::NotFound
      keep[i] = it->second;
      qs[i] = reqs[i] + UUID_V4_STR_BYTES;
    } else {
      // bin/server.rs:123-138: the public parameters travel with the query
      if (lens[i] != S.setup_bytes + S.query_bytes)
        throw Fail{SP_E_ARG, "request " + std::to_string(i) + ": expected setup_bytes + query_bytes"};
      keep[i] = deserialize_pp(S, reqs[i], S.setup_bytes);
      qs[i] = reqs[i] + S.setup_bytes;
    }
    pps[i] = keep[i].get();
  }
  size_t one = 0;
  const int rc = sp_process_query_batch(S.params, pps.data(), qs.data(), qlens.data(), n, S.db, out, stride, &one);
  if (rc != SP_OK) throw Fail{rc, std::string("sp_process_query_batch: ") + sp_last_error()};
  for (int i = 0; i < n; i++) out_lens[i] = one;
}

extern "C" {

sp_server_t* sp_server_create(const sp_params_t* params, const sp_db_t* db) {
  sp_server_t* out = nullptr;
  const int rc = guarded_ep([&] {
    if (!params || !db) throw Fail{SP_E_ARG, "null argument"};
    auto s = std::make_unique<sp_server>();
    s->params = params;
    s->db = db;
    s->setup_bytes = (size_t)sp_params_get(params, "setup_bytes");
    s->query_bytes = (size_t)sp_params_get(params, "query_bytes");
    s->response_bytes = (size_t)sp_params_get(params, "response_bytes");
    s->expand_queries = sp_params_get(params, "expand_queries") != 0;
    out = s.release();
  });
  return rc == SP_OK ? out : nullptr;
}

void sp_server_free(sp_server_t* s) { delete s; }

size_t sp_server_clients(const sp_server_t* s) {
  if (!s) return 0;
  std::shared_lock<std::shared_mutex> lk(s->mu);
  return s->pub_params.size();
}

int sp_server_setup(sp_server_t* s, const uint8_t* pp_bytes, size_t len, char* uuid_out37) {
  return guarded_ep([&] {
    if (!s || !pp_bytes || !uuid_out37) throw Fail{SP_E_ARG, "null argument"};
    auto pp = deserialize_pp(*s, pp_bytes, len);
    const std::string id = uuid_v4();
    {
      std::unique_lock<std::shared_mutex> lk(s->mu);
      s->pub_params[id] = std::move(pp);
    }
    memcpy(uuid_out37, id.c_str(), UUID_V4_STR_BYTES + 1);
  });
}

int sp_server_forget(sp_server_t* s, const char* uuid) {
  return guarded_ep([&] {
    if (!s || !uuid) throw Fail{SP_E_ARG, "null argument"};
    std::unique_lock<std::shared_mutex> lk(s->mu);
    if (s->pub_params.erase(uuid) == 0) throw Fail{SP_E_NOTFOUND, std::string("unknown uuid ") + uuid};
  });
}

int sp_server_setup_json(sp_server_t* s, const char* body, size_t body_len, char* out, size_t out_cap, size_t* out_len) {
  return guarded_ep([&] {
    if (!s || !body || !out || !out_len) throw Fail{SP_E_ARG, "null argument"};
    const char* p = body;
    std::string b64;
    if (!parse_string(p, body + body_len, b64)) throw Fail{SP_E_ARG, "/setup body is not a JSON string"};
    skip_ws(p, body + body_len);
    if (p != body + body_len) throw Fail{SP_E_ARG, "/setup body: trailing data"};
    std::vector<uint8_t> raw;
    if (!b64_decode(b64.data(), b64.size(), raw)) throw Fail{SP_E_ARG, "/setup body is not valid base64"};
    char id[UUID_V4_STR_BYTES + 1];
    auto pp = deserialize_pp(*s, raw.data(), raw.size());
    const std::string u = uuid_v4();
    {
      std::unique_lock<std::shared_mutex> lk(s->mu);
      s->pub_params[u] = std::move(pp);
    }
    memcpy(id, u.c_str(), UUID_V4_STR_BYTES + 1);
    const std::string resp = std::string("{\"uuid\":\"") + id + "\"}";  // serde_json::to_string(&UuidResponse { uuid })
    if (resp.size() + 1 > out_cap) throw Fail{SP_E_ARG, "output buffer too small"};
    memcpy(out, resp.c_str(), resp.size() + 1);
    *out_len = resp.size();
  });
}

int sp_server_private_read(sp_server_t* s, const uint8_t* const* requests, const size_t* request_lens, int n, uint8_t* out,
                           size_t out_stride, size_t* out_lens) {
  return guarded_ep([&] {
    if (!s || n < 0 || (n > 0 && (!requests || !request_lens || !out || !out_lens))) throw Fail{SP_E_ARG, "null argument"};
    private_read(*s, requests, request_lens, n, out, out_stride, out_lens);
  });
}

size_t sp_server_private_read_json_bound(const sp_server_t* s, int n_queries) {
  if (!s || n_queries < 0) return 0;
  return 2 + (size_t)n_queries * ((s->response_bytes + 2) / 3 * 4 + 3) + 1;
}

int sp_server_private_read_json(sp_server_t* s, const char* body, size_t body_len, char* out, size_t out_cap, size_t* out_len) {
  return guarded_ep([&] {
    if (!s || !body || !out || !out_len) throw Fail{SP_E_ARG, "null argument"};
    std::vector<std::string> items;
    if (!parse_string_list(body, body_len, items)) throw Fail{SP_E_ARG, "/private-read body is not a JSON list of strings"};
    const int n = (int)items.size();
    std::vector<std::vector<uint8_t>> raw((size_t)n);
    std::vector<const uint8_t*> reqs((size_t)n);
    std::vector<size_t> lens((size_t)n);
    for (int i = 0; i < n; i++) {
      if (!b64_decode(items[i].data(), items[i].size(), raw[i])) throw Fail{SP_E_ARG, "request " + std::to_string(i) + " is not valid base64"};
      reqs[i] = raw[i].data();
      lens[i] = raw[i].size();
    }
    std::vector<uint8_t> resp((size_t)n * s->response_bytes);
    std::vector<size_t> rl((size_t)n, 0);
    private_read(*s, reqs.data(), lens.data(), n, resp.data(), s->response_bytes, rl.data());
    std::string js = "[";  // serde_json::to_string(&Vec<String>)
    for (int i = 0; i < n; i++) {
      if (i) js += ',';
      js += '"';
      js += b64_encode(resp.data() + (size_t)i * s->response_bytes, rl[i]);
      js += '"';
    }
    js += ']';
    if (js.size() + 1 > out_cap) throw Fail{SP_E_ARG, "output buffer too small (see sp_server_private_read_json_bound)"};
    memcpy(out, js.c_str(), js.size() + 1);
    *out_len = js.size();
  });
}

}  // extern "C"
