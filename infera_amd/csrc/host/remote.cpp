#include "remote.hpp"

#include <arpa/inet.h>
#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <netdb.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>
#include <utime.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <thread>
#include <vector>

#include "common.hpp"

namespace infera_hip::remote {

namespace {

// ---- SHA-256 (FIPS 180-4) ----------------------------------------------------------------------------
constexpr uint32_t kK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

void sha256_block(uint32_t h[8], const uint8_t *p) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = uint32_t(p[4 * i]) << 24 | uint32_t(p[4 * i + 1]) << 16 | uint32_t(p[4 * i + 2]) << 8 | p[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; i++) {
    const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + kK[i] + w[i];
    const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), maj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + maj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// ---- small filesystem helpers ------------------------------------------------------------------------------
bool exists(const std::string &p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0;
}
bool ends_with(const std::string &s, const std::string &suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}
void mkdir_p(const std::string &dir) {
  std::string cur;
  for (size_t i = 0; i <= dir.size(); i++) {
    if (i == dir.size() || dir[i] == '/') {
      if (!cur.empty() && !exists(cur) && ::mkdir(cur.c_str(), 0755) != 0 && errno != EEXIST) throw InferaError::cache_dir(std::strerror(errno));
    }
    if (i < dir.size()) cur += dir[i];
  }
}
// http.rs:56-62: refresh the access time so the LRU order sees this use
void touch_atime(const std::string &path) {
  struct stat st;
  if (::stat(path.c_str(), &st) != 0) return;
  struct utimbuf t;
  t.actime = ::time(nullptr);
  t.modtime = st.st_mtime;
  if (::utime(path.c_str(), &t) != 0) throw InferaError::io(std::strerror(errno));
}

struct CachedFile {
  std::string path;
  struct timespec atime;
  uint64_t size;
};
// http.rs:65-89: *.onnx files of the cache directory, least recently accessed first
std::vector<CachedFile> cached_by_atime(const std::string &dir) {
  std::vector<CachedFile> out;
  DIR *d = ::opendir(dir.c_str());
  if (!d) return out;
  while (dirent *e = ::readdir(d)) {
    const std::string p = dir + "/" + e->d_name;
    struct stat st;
    if (ends_with(p, ".onnx") && ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode)) out.push_back({p, st.st_atim, uint64_t(st.st_size)});
  }
  ::closedir(d);
  std::sort(out.begin(), out.end(), [](const CachedFile &a, const CachedFile &b) {
    return a.atime.tv_sec != b.atime.tv_sec ? a.atime.tv_sec < b.atime.tv_sec : a.atime.tv_nsec < b.atime.tv_nsec;
  });
  return out;
}
// http.rs:98-120
void evict_if_needed(const std::string &dir, uint64_t required) {
  const uint64_t limit = Config::get().cache_size_limit;
  auto files = cached_by_atime(dir);
  uint64_t current = 0;
  for (const auto &f : files) current += f.size;
  if (current + required <= limit) return;
  const uint64_t target = limit > required ? limit - required : 0;
  uint64_t freed = 0;
  for (const auto &f : files) {
    if (current - freed <= target) break;
    if (::unlink(f.path.c_str()) != 0) throw InferaError::io(std::strerror(errno));
    ::unlink((f.path.substr(0, f.path.size() - 5) + ".etag").c_str());  // its revalidation tag goes with it
    freed += f.size;
  }
}

// Deletes the partial download unless committed (http.rs:15-43).
struct TempFileGuard {
  std::string path;
  bool committed = false;
  ~TempFileGuard() {
    if (!committed) ::unlink(path.c_str());
  }
};

// ---- minimal HTTP/1.1 client ---------------------------------------------------------------------------------
struct Url {
  std::string host, port, path;
};
Url parse_url(const std::string &url) {
  if (url.rfind("https://", 0) == 0) throw InferaError::http("https is not supported by the built-in client (INFERA_HTTP_BACKEND=socket): " + url);
  if (url.rfind("http://", 0) != 0) throw InferaError::http("builder error: relative URL without a base: " + url);
  Url u;
  const size_t hs = 7, slash = url.find('/', hs);
  std::string hostport = url.substr(hs, slash == std::string::npos ? std::string::npos : slash - hs);
  if (const size_t frag = hostport.find('#'); frag != std::string::npos) hostport.resize(frag);
  u.path = slash == std::string::npos ? "/" : url.substr(slash);
  if (const size_t frag = u.path.find('#'); frag != std::string::npos) u.path.resize(frag);  // a fragment is never sent (RFC 9110 7.1)
  if (u.path.empty()) u.path = "/";
  const size_t at = hostport.rfind('@');
  if (at != std::string::npos) hostport = hostport.substr(at + 1);
  const size_t colon = hostport.rfind(':');
  if (colon != std::string::npos && hostport.find(']') == std::string::npos) {
    u.host = hostport.substr(0, colon);
    u.port = hostport.substr(colon + 1);
  } else {
    u.host = hostport;
    u.port = "80";
  }
  if (u.host.empty()) throw InferaError::http("builder error: empty host: " + url);
  return u;
}

using Clock = std::chrono::steady_clock;

struct Conn {
  int fd = -1;
  Clock::time_point deadline;
  std::string buf;  // bytes received but not consumed yet
  ~Conn() {
    if (fd >= 0) ::close(fd);
  }
  void arm_timeout() {
    const auto left = std::chrono::duration_cast<std::chrono::microseconds>(deadline - Clock::now()).count();
    if (left <= 0) throw InferaError::http("operation timed out");
    struct timeval tv;
    tv.tv_sec = time_t(left / 1000000);
    tv.tv_usec = suseconds_t(left % 1000000);
    ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    ::setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
  }
  // false on orderly close
  bool fill() {
    arm_timeout();
    char tmp[16384];
    const ssize_t n = ::recv(fd, tmp, sizeof tmp, 0);
    if (n < 0) {
      if (errno == EAGAIN || errno == EWOULDBLOCK) throw InferaError::http("operation timed out");
      throw InferaError::http(std::string("error reading a body from connection: ") + std::strerror(errno));
    }
    if (n == 0) return false;
    buf.append(tmp, size_t(n));
    return true;
  }
  void send_all(const std::string &s) {
    size_t off = 0;
    while (off < s.size()) {
      arm_timeout();
      const ssize_t n = ::send(fd, s.data() + off, s.size() - off, MSG_NOSIGNAL);
      if (n <= 0) throw InferaError::http(std::string("error sending request: ") + std::strerror(errno));
      off += size_t(n);
    }
  }
  std::string read_line() {
    for (;;) {
      const size_t nl = buf.find('\n');
      if (nl != std::string::npos) {
        std::string line = buf.substr(0, nl);
        buf.erase(0, nl + 1);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return line;
      }
      if (!fill()) throw InferaError::http("connection closed before message completed");
    }
  }
};

void connect_to(Conn &c, const Url &u) {
  struct addrinfo hints {};
  hints.ai_family = AF_UNSPEC;
  hints.ai_socktype = SOCK_STREAM;
  struct addrinfo *res = nullptr;
  const int rc = ::getaddrinfo(u.host.c_str(), u.port.c_str(), &hints, &res);
  if (rc != 0) throw InferaError::http("error sending request: dns error: " + std::string(::gai_strerror(rc)));
  std::string last = "no address";
  for (struct addrinfo *a = res; a; a = a->ai_next) {
    const int fd = ::socket(a->ai_family, a->ai_socktype, a->ai_protocol);
    if (fd < 0) continue;
    c.fd = fd;
    try {
      c.arm_timeout();
    } catch (...) {
      ::freeaddrinfo(res);
      throw;
    }
    if (::connect(fd, a->ai_addr, a->ai_addrlen) == 0) {
      ::freeaddrinfo(res);
      return;
    }
    last = std::strerror(errno);
    ::close(fd);
    c.fd = -1;
  }
  ::freeaddrinfo(res);
  throw InferaError::http("error sending request: tcp connect error: " + last);
}

std::string lower(std::string s) {
  for (auto &ch : s) ch = char(::tolower(static_cast<unsigned char>(ch)));
  return s;
}

enum class Fetch { NotModified, Downloaded };

// http.rs:303-337 (download_file): GET with optional If-None-Match; 304 -> NotModified; other non-2xx -> error;
// body streamed to `dest`.  `etag_out` receives the response's ETag ("" if none).
Fetch download_file_curl(const std::string &url, const std::string &dest, uint64_t timeout_secs, const std::string *etag, std::string &etag_out);

Fetch download_file(const std::string &url_in, const std::string &dest, uint64_t timeout_secs, const std::string *etag, std::string &etag_out) {
  std::string url = url_in;
  const auto deadline = Clock::now() + std::chrono::seconds(timeout_secs ? timeout_secs : 1);
  for (int hop = 0; hop < 10; hop++) {
    const Url u = parse_url(url);
    Conn c;
    c.deadline = deadline;
    connect_to(c, u);
    std::string req = "GET " + u.path + " HTTP/1.1\r\nHost: " + u.host + (u.port == "80" ? "" : ":" + u.port) +
                      "\r\nUser-Agent: infera-mi355x\r\nAccept: */*\r\nConnection: close\r\n";
    if (etag) req += "If-None-Match: " + *etag + "\r\n";
    c.send_all(req + "\r\n");

    const std::string status_line = c.read_line();
    int status = 0;
    if (std::sscanf(status_line.c_str(), "HTTP/%*d.%*d %d", &status) != 1) throw InferaError::http("invalid HTTP response: " + status_line);
    int64_t content_length = -1;
    bool chunked = false;
    std::string location;
    etag_out.clear();
    for (;;) {
      const std::string line = c.read_line();
      if (line.empty()) break;
      const size_t colon = line.find(':');
      if (colon == std::string::npos) continue;
      const std::string key = lower(line.substr(0, colon));
      std::string val = line.substr(colon + 1);
      val.erase(0, val.find_first_not_of(" \t"));
      val.erase(val.find_last_not_of(" \t") + 1);
      if (key == "content-length") content_length = std::strtoll(val.c_str(), nullptr, 10);
      else if (key == "transfer-encoding" && lower(val).find("chunked") != std::string::npos) chunked = true;
      else if (key == "etag") etag_out = val;
      else if (key == "location") location = val;
    }
    if (status == 304) return Fetch::NotModified;
    if (status >= 300 && status < 400 && !location.empty()) {  // reqwest follows redirects
      url = location.rfind("http", 0) == 0 ? location : "http://" + u.host + (u.port == "80" ? "" : ":" + u.port) + location;
      // a hop to https leaves this client: hand the rest of the chain to the TLS-capable backend
      if (url.rfind("https://", 0) == 0) return download_file_curl(url, dest, timeout_secs, etag, etag_out);
      continue;
    }
    if (status < 200 || status >= 300) {
      const size_t sp = status_line.find(' ');
      throw InferaError::http("HTTP status " + std::string(status >= 500 ? "server" : "client") + " error (" +
                              (sp == std::string::npos ? std::to_string(status) : status_line.substr(sp + 1)) + ") for url (" + url + ")");
    }

    FILE *fp = std::fopen(dest.c_str(), "wb");
    if (!fp) throw InferaError::io(std::strerror(errno));
    struct Closer {
      FILE *f;
      ~Closer() { std::fclose(f); }
    } closer{fp};
    auto put = [&](const char *p, size_t n) {
      if (n && std::fwrite(p, 1, n, fp) != n) throw InferaError::io(std::strerror(errno));
    };
    if (chunked) {
      for (;;) {
        // chunk-size line: 1..16 hex digits, optionally followed by ";extensions".  Anything else (an empty line
        // where the connection died, garbage) must not read as "0 = end of body" and commit a truncated file.
        const std::string szl = c.read_line();
        size_t nd = 0;
        while (nd < szl.size() && std::isxdigit(static_cast<unsigned char>(szl[nd]))) nd++;
        size_t rest = nd;
        while (rest < szl.size() && (szl[rest] == ' ' || szl[rest] == '\t')) rest++;
        if (nd == 0 || nd > 16 || (rest < szl.size() && szl[rest] != ';')) throw InferaError::io("unexpected end of file");
        const uint64_t n = std::strtoull(szl.substr(0, nd).c_str(), nullptr, 16);
        if (n == 0) break;
        uint64_t left = n;
        while (left) {
          if (c.buf.empty() && !c.fill()) throw InferaError::io("unexpected end of file");
          const size_t take = size_t(std::min<uint64_t>(left, c.buf.size()));
          put(c.buf.data(), take);
          c.buf.erase(0, take);
          left -= take;
        }
        (void)c.read_line();  // CRLF after the chunk
      }
    } else if (content_length >= 0) {
      uint64_t left = uint64_t(content_length);
      while (left) {
        // the server promised `content_length` bytes: a short body is an error (http.rs tests :346-378, :416-459)
        if (c.buf.empty() && !c.fill()) throw InferaError::io("unexpected end of file");
        const size_t take = size_t(std::min<uint64_t>(left, c.buf.size()));
        put(c.buf.data(), take);
        c.buf.erase(0, take);
        left -= take;
      }
    } else {
      do {
        put(c.buf.data(), c.buf.size());
        c.buf.clear();
      } while (c.fill());
    }
    if (std::fflush(fp) != 0) throw InferaError::io(std::strerror(errno));
    return Fetch::Downloaded;
  }
  throw InferaError::http("error following redirect: too many redirects for url (" + url_in + ")");
}

// ---- libcurl backend (https, or everything when INFERA_HTTP_BACKEND=curl) ----------------------------------------
// The image ships libcurl.so.4 but no TLS development files; the handful of easy-interface entry points is resolved
// with dlopen at first use (ABI-stable since 7.x), so the library has no link-time dependency and an absent libcurl
// only means https:// stays unsupported.
struct CurlApi {
  void *(*easy_init)() = nullptr;
  int (*easy_setopt)(void *, int, ...) = nullptr;
  int (*easy_perform)(void *) = nullptr;
  int (*easy_getinfo)(void *, int, ...) = nullptr;
  void (*easy_cleanup)(void *) = nullptr;
  const char *(*easy_strerror)(int) = nullptr;
  void *(*slist_append)(void *, const char *) = nullptr;
  void (*slist_free_all)(void *) = nullptr;
  bool ok = false;
};
const CurlApi &curl_api() {
  static const CurlApi api = [] {
    CurlApi a;
    void *h = nullptr;
    for (const char *name : {"libcurl.so.4", "libcurl-gnutls.so.4", "libcurl.so"})
      if ((h = ::dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return a;
    auto sym = [&](const char *n) { return ::dlsym(h, n); };
    a.easy_init = reinterpret_cast<void *(*)()>(sym("curl_easy_init"));
    a.easy_setopt = reinterpret_cast<int (*)(void *, int, ...)>(sym("curl_easy_setopt"));
    a.easy_perform = reinterpret_cast<int (*)(void *)>(sym("curl_easy_perform"));
    a.easy_getinfo = reinterpret_cast<int (*)(void *, int, ...)>(sym("curl_easy_getinfo"));
    a.easy_cleanup = reinterpret_cast<void (*)(void *)>(sym("curl_easy_cleanup"));
    a.easy_strerror = reinterpret_cast<const char *(*)(int)>(sym("curl_easy_strerror"));
    a.slist_append = reinterpret_cast<void *(*)(void *, const char *)>(sym("curl_slist_append"));
    a.slist_free_all = reinterpret_cast<void (*)(void *)>(sym("curl_slist_free_all"));
    a.ok = a.easy_init && a.easy_setopt && a.easy_perform && a.easy_getinfo && a.easy_cleanup && a.easy_strerror && a.slist_append &&
           a.slist_free_all;
    return a;
  }();
  return api;
}

struct CurlSink {
  FILE *fp = nullptr;
  std::string etag;
  bool write_failed = false;
};
size_t curl_write_cb(char *p, size_t sz, size_t n, void *ud) {
  auto *s = static_cast<CurlSink *>(ud);
  if (std::fwrite(p, sz, n, s->fp) != n) {
    s->write_failed = true;
    return 0;
  }
  return sz * n;
}
size_t curl_header_cb(char *p, size_t sz, size_t n, void *ud) {
  auto *s = static_cast<CurlSink *>(ud);
  std::string line(p, sz * n);
  const size_t colon = line.find(':');
  if (colon != std::string::npos && lower(line.substr(0, colon)) == "etag") {
    std::string v = line.substr(colon + 1);
    v.erase(0, v.find_first_not_of(" \t"));
    v.erase(v.find_last_not_of(" \t\r\n") + 1);
    s->etag = v;
  }
  return sz * n;
}

Fetch download_file_curl(const std::string &url, const std::string &dest, uint64_t timeout_secs, const std::string *etag, std::string &etag_out) {
  const CurlApi &c = curl_api();
  if (!c.ok) throw InferaError::http("https is not supported by this build (no TLS library, libcurl not loadable): " + url);
  // option / info codes of curl/curl.h (stable ABI values)
  enum { kUrl = 10002, kWriteFn = 20011, kWriteData = 10001, kHeaderFn = 20079, kHeaderData = 10029, kFollow = 52, kTimeout = 13,
         kHttpHeader = 10023, kNoSignal = 99, kUserAgent = 10018, kMaxRedirs = 68, kInfoResponseCode = 0x200002, kPartialFile = 18,
         kTimedOut = 28 };
  void *h = c.easy_init();
  if (!h) throw InferaError::http("curl_easy_init failed");
  CurlSink sink;
  sink.fp = std::fopen(dest.c_str(), "wb");
  if (!sink.fp) {
    c.easy_cleanup(h);
    throw InferaError::io(std::strerror(errno));
  }
  void *hdrs = nullptr;
  if (etag) hdrs = c.slist_append(hdrs, ("If-None-Match: " + *etag).c_str());
  c.easy_setopt(h, kUrl, url.c_str());
  c.easy_setopt(h, kWriteFn, curl_write_cb);
  c.easy_setopt(h, kWriteData, &sink);
  c.easy_setopt(h, kHeaderFn, curl_header_cb);
  c.easy_setopt(h, kHeaderData, &sink);
  c.easy_setopt(h, kFollow, 1L);
  c.easy_setopt(h, kMaxRedirs, 10L);
  c.easy_setopt(h, kTimeout, long(timeout_secs ? timeout_secs : 1));
  c.easy_setopt(h, kNoSignal, 1L);
  c.easy_setopt(h, kUserAgent, "infera-mi355x");
  if (hdrs) c.easy_setopt(h, kHttpHeader, hdrs);
  const int rc = c.easy_perform(h);
  long status = 0;
  c.easy_getinfo(h, kInfoResponseCode, &status);
  const bool flush_ok = std::fclose(sink.fp) == 0;
  if (hdrs) c.slist_free_all(hdrs);
  c.easy_cleanup(h);
  if (sink.write_failed || (!flush_ok && rc == 0)) throw InferaError::io("cannot write " + dest);
  if (rc == kPartialFile) throw InferaError::io("unexpected end of file");
  if (rc == kTimedOut) throw InferaError::http("operation timed out");
  if (rc != 0) throw InferaError::http(std::string("error sending request: ") + c.easy_strerror(rc));
  if (status == 304) return Fetch::NotModified;
  if (status < 200 || status >= 300)
    throw InferaError::http("HTTP status " + std::string(status >= 500 ? "server" : "client") + " error (" + std::to_string(status) + ") for url (" + url + ")");
  etag_out = sink.etag;
  return Fetch::Downloaded;
}

// http for the built-in client, https (or INFERA_HTTP_BACKEND=curl) for libcurl
Fetch fetch(const std::string &url, const std::string &dest, uint64_t timeout_secs, const std::string *etag, std::string &etag_out) {
  static const std::string backend = [] {
    const char *v = std::getenv("INFERA_HTTP_BACKEND");
    return std::string(v ? v : "auto");
  }();
  const bool https = url.rfind("https://", 0) == 0;
  if (backend == "curl" || (backend != "socket" && https)) return download_file_curl(url, dest, timeout_secs, etag, etag_out);
  return download_file(url, dest, timeout_secs, etag, etag_out);
}

std::string read_trimmed(const std::string &path) {
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) return "";
  std::string s;
  char tmp[256];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof tmp, fp)) > 0) s.append(tmp, n);
  std::fclose(fp);
  const size_t b = s.find_first_not_of(" \t\r\n"), e = s.find_last_not_of(" \t\r\n");
  return b == std::string::npos ? "" : s.substr(b, e - b + 1);
}

}  // namespace

std::string sha256_hex(const std::string &data) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  std::string m = data;
  const uint64_t bits = uint64_t(data.size()) * 8;
  m.push_back(char(0x80));
  while (m.size() % 64 != 56) m.push_back(0);
  for (int i = 7; i >= 0; i--) m.push_back(char((bits >> (8 * i)) & 0xff));
  for (size_t off = 0; off < m.size(); off += 64) sha256_block(h, reinterpret_cast<const uint8_t *>(m.data()) + off);
  char hex[65];
  for (int i = 0; i < 8; i++) std::snprintf(hex + 8 * i, 9, "%08x", h[i]);
  return std::string(hex, 64);
}

std::string handle_remote_model(const std::string &url) {
  const Config &cfg = Config::get();
  const std::string dir = cfg.cache_dir;
  if (!exists(dir)) {
    log_msg(2, "Creating cache directory: " + dir);
    mkdir_p(dir);
  }
  const std::string key = sha256_hex(url);
  const std::string cached = dir + "/" + key + ".onnx", etag_path = dir + "/" + key + ".etag";

  std::string local_etag;
  bool have_etag = false;
  if (exists(cached)) {
    if (exists(etag_path)) {
      local_etag = read_trimmed(etag_path);
      have_etag = true;
      log_msg(2, "Found local ETag metadata for URL: " + url);
    } else {
      // no validator stored: trust the cached copy without a request (http.rs:200-208)
      log_msg(2, "Cache hit for URL (no ETag metadata): " + url);
      touch_atime(cached);
      return cached;
    }
  }

  TempFileGuard guard{dir + "/" + key + ".onnx.part"};
  const uint32_t attempts = std::max<uint32_t>(1, cfg.http_retry_attempts);
  InferaError last = InferaError::http("Unknown error");
  for (uint32_t attempt = 1; attempt <= attempts; attempt++) {
    log_msg(3, "Download/Validation attempt " + std::to_string(attempt) + "/" + std::to_string(attempts) + " for " + url);
    try {
      std::string new_etag;
      const Fetch r = fetch(url, guard.path, cfg.http_timeout_secs, have_etag ? &local_etag : nullptr, new_etag);
      if (r == Fetch::NotModified) {
        log_msg(2, "Cache hit (ETag verified) for URL: " + url);
        touch_atime(cached);
        return cached;
      }
      log_msg(2, "Successfully downloaded: " + url);
      struct stat st;
      if (::stat(guard.path.c_str(), &st) != 0) throw InferaError::io(std::strerror(errno));
      evict_if_needed(dir, uint64_t(st.st_size));
      if (::rename(guard.path.c_str(), cached.c_str()) != 0) throw InferaError::io(std::strerror(errno));
      if (!new_etag.empty()) {
        FILE *fp = std::fopen(etag_path.c_str(), "wb");
        if (fp) {
          std::fwrite(new_etag.data(), 1, new_etag.size(), fp);
          std::fclose(fp);
        } else {
          log_msg(1, std::string("Failed to write ETag metadata: ") + std::strerror(errno));
        }
      } else {
        ::unlink(etag_path.c_str());
      }
      guard.committed = true;
      return cached;
    } catch (const InferaError &e) {
      log_msg(1, "Download/Validation attempt " + std::to_string(attempt) + "/" + std::to_string(attempts) + " failed: " + e.what());
      last = e;
      ::unlink(guard.path.c_str());
      if (attempt < attempts) std::this_thread::sleep_for(std::chrono::milliseconds(cfg.http_retry_delay_ms * attempt));
    }
  }
  log_msg(0, "Failed to download/validate after " + std::to_string(attempts) + " attempts: " + url);
  throw last;
}

}  // namespace infera_hip::remote
