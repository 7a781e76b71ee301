// Error sink for the sanitizer build of the host-only readers (tests/test_host_sanitized_cpu.py): the product's own
// q3_set_err / q3tts_last_error live in q3_engine.hip, which needs HIP.
#include <string>
struct q3tts_engine;
static thread_local std::string g_err;
int q3_set_err(q3tts_engine*, int code, const std::string& msg) { g_err = msg; return code; }
extern "C" const char* q3tts_last_error(const q3tts_engine*) { return g_err.c_str(); }
