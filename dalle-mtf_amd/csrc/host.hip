// host.hip -- host-side helpers of the input format: no device code.
// TFRecord framing protects the length and the payload of every record with a masked CRC-32C
// (tensorflow/core/lib/io/record_reader.cc, read through tf.data.TFRecordDataset at the reference's src/input_fns.py:111).
// The reader verifies them as TensorFlow does; at ~1900 records/s/GPU (JPEG payloads of tens of KB) that is a host cost
// worth a table-driven loop rather than the interpreter: slicing-by-8, 8 bytes per iteration, ~2 GB/s per thread, no GIL
// (ctypes releases it for the call).
#include "common.h"
#include <stddef.h>
#include <string.h>

static uint32_t g_crc_tab[8][256];
static int g_crc_ready = 0;

static void crc_init() {
  for (int i = 0; i < 256; ++i) {
    uint32_t c = (uint32_t)i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);   // Castagnoli, reflected
    g_crc_tab[0][i] = c;
  }
  for (int i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xFF];
  __atomic_store_n(&g_crc_ready, 1, __ATOMIC_RELEASE);
}

// CRC-32C (init 0xFFFFFFFF, final xor) of data[0..n).  Pure host function; n == 0 -> 0.
extern "C" uint32_t dmi_crc32c(const void* data, size_t n) {
  if (!__atomic_load_n(&g_crc_ready, __ATOMIC_ACQUIRE)) crc_init();   // idempotent: racing initialisers write equal values
  const uint8_t* p = (const uint8_t*)data;
  uint32_t c = 0xFFFFFFFFu;
  while (n && ((uintptr_t)p & 7)) { c = g_crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8); --n; }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = g_crc_tab[7][w & 0xFF] ^ g_crc_tab[6][(w >> 8) & 0xFF] ^ g_crc_tab[5][(w >> 16) & 0xFF] ^
        g_crc_tab[4][(w >> 24) & 0xFF] ^ g_crc_tab[3][(w >> 32) & 0xFF] ^ g_crc_tab[2][(w >> 40) & 0xFF] ^
        g_crc_tab[1][(w >> 48) & 0xFF] ^ g_crc_tab[0][(w >> 56) & 0xFF];
    p += 8; n -= 8;
  }
  while (n--) c = g_crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
