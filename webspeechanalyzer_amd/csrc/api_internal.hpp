// api_internal.hpp — what api.hip (batches) and stream_api.hip (streams) share: the context object and
// the error plumbing of the C ABI.
#pragma once
#include <string>
#include "wsa_internal.hpp"

struct wsa_ctx {
    wsa_config cfg;
    int device = 0;
    int n_cu = 0;
    std::string err;
};

namespace wsa_api {
extern thread_local std::string g_create_error;
inline wsa_status fail(wsa_ctx* c, wsa_status st, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return st;
}
}  // namespace wsa_api

#define HIP_TRY(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
        return wsa_api::fail((ctx), WSA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
