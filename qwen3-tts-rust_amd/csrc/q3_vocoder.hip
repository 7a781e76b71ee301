// placeholder until the vocoder lands (next commit): engine runs codes-only
#include "q3_engine.h"
int q3_voc_create(q3tts_engine* e) { e->voc = nullptr; return Q3TTS_OK; }
void q3_voc_destroy(q3tts_engine*) {}
int q3_voc_reset(q3tts_engine*, int) { return Q3TTS_OK; }
int q3_voc_decode(q3tts_engine* e, int, int, int, int, hipStream_t) { return q3_set_err(e, Q3TTS_ERR_UNSUPPORTED, "vocoder not built"); }
float* q3_voc_pcm(q3tts_engine*, int) { return nullptr; }
int q3_voc_samples(q3tts_engine*, int) { return 0; }
int q3_voc_samples_per_frame(const q3tts_engine*) { return 1920; }
extern "C" int q3tts_k_vocoder(q3tts_engine* e, const int32_t*, int32_t, int32_t, float*, int32_t*) { return q3_set_err(e, Q3TTS_ERR_UNSUPPORTED, "vocoder not built"); }
