// remote.hpp -- remote model fetch + LRU disk cache (reference: infera/src/http.rs).
//
// infera_load_model("name", "http://...") downloads the model into the cache directory
// (INFERA_CACHE_DIR, default $TMPDIR/infera_cache) under sha256(url).onnx, revalidates it with the stored
// ETag (If-None-Match -> 304) on later loads, evicts least-recently-accessed files beyond
// INFERA_CACHE_SIZE_LIMIT, retries with linear back-off and never leaves a partial file behind.
// http:// goes through a built-in HTTP/1.1 client over POSIX sockets (Content-Length, chunked and read-to-close
// bodies, redirects); https:// through libcurl, resolved with dlopen at first use (no link-time dependency; if it
// cannot be loaded https:// fails with "HTTP request failed: ...").  INFERA_HTTP_BACKEND=socket|curl forces one.
#pragma once

#include <string>

namespace infera_hip::remote {

// http.rs:179-300.  Returns the path of the cached file; throws InferaError (http / io / cache_dir).
std::string handle_remote_model(const std::string &url);

// sha256(url) as lower-case hex: the cache key (http.rs:186-190).  Exposed for tests.
std::string sha256_hex(const std::string &data);

}  // namespace infera_hip::remote
