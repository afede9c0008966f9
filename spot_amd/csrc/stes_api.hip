// Version and error reporting shared by every entry point of libstes_hip.
#include "stes_common.h"
#include <string.h>

static thread_local char g_err[256] = "";

int stes_set_error(int code, const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
    return code;
}

extern "C" int stes_version(void) { return 1; }
extern "C" const char* stes_last_error(void) { return g_err; }

// ---------------------------------------------------------------- deterministic mode
// Process-wide: the ticket words and the running turn counter.  Launches that commit through the ticket must reach the GPU
// in the order in which their windows were taken, i.e. the caller drives one stream while the mode is on.
#include <mutex>
static std::mutex g_det_mu;
static unsigned* g_det_ticket = nullptr;
static unsigned g_det_next = 0;

extern "C" int stes_set_deterministic(uint32_t* ticket, void* stream) {
    std::lock_guard<std::mutex> lock(g_det_mu);
    if (ticket) {
        // both words start at zero; the turn counter then only ever grows (modulo 2^32, compared for equality)
        if (hipMemsetAsync(ticket, 0, 256, (hipStream_t)stream) != hipSuccess) return stes_set_error(STES_ERR_LAUNCH, "stes_set_deterministic: cannot clear the ticket");
    }
    g_det_ticket = ticket;
    g_det_next = 0;
    return STES_OK;
}

StesDet stes_det_take(unsigned nblocks) {
    std::lock_guard<std::mutex> lock(g_det_mu);
    StesDet d{g_det_ticket, g_det_next, 0u};
    if (g_det_ticket) g_det_next += nblocks;
    return d;
}
