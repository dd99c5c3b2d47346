// Host-side restatement of the reference's batch-verification transcript
// (ed25519-dalek/src/batch.rs:168-222 over batch/transcript.rs:39-207): Merlin framing on
// STROBE-128/1600 (strobe-rs 0.13.0 / keccak 0.2.0 in the reference's Cargo.lock; restated from the
// public STROBE v1.0.2 and FIPS 202 specifications).  It is inherently sequential (one sponge absorbs
// 2n messages and is then squeezed n times), so z_mode 0 runs it on one host core; z_mode 1 replaces
// it with a parallel on-device derivation.
#pragma once
#include <stdint.h>
#include <string.h>
#include "constants_gen.h"

namespace c25519_tr {

static inline uint64_t rotl(uint64_t x, unsigned n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static inline void keccak_f(uint64_t a[25]) {
    static const uint64_t RC[24] = C25519_KECCAK_RC;
    static const unsigned ROT[25] = C25519_KECCAK_ROT;
    for (int rnd = 0; rnd < 24; rnd++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) { uint64_t d = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1); for (int y = 0; y < 25; y += 5) a[x + y] ^= d; }
        for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 25; y += 5) for (int x = 0; x < 5; x++) a[x + y] = b[x + y] ^ (~b[(x + 1) % 5 + y] & b[(x + 2) % 5 + y]);
        a[0] ^= RC[rnd];
    }
}

struct strobe {
    enum { R = 166, FI = 1, FA = 2, FC = 4, FT = 8, FM = 16, FK = 32 };
    uint64_t st[25]; uint8_t pos, pos_begin;
    uint8_t *bytes() { return (uint8_t *)st; }
    void run_f() { uint8_t *b = bytes(); b[pos] ^= pos_begin; b[pos + 1] ^= 0x04; b[R + 1] ^= 0x80; keccak_f(st); pos = 0; pos_begin = 0; }
    void absorb(const uint8_t *d, size_t n) { uint8_t *b = bytes(); for (size_t i = 0; i < n; i++) { b[pos] ^= d[i]; if (++pos == R) run_f(); } }
    void overwrite(const uint8_t *d, size_t n) { uint8_t *b = bytes(); for (size_t i = 0; i < n; i++) { b[pos] = d[i]; if (++pos == R) run_f(); } }
    void squeeze(uint8_t *d, size_t n) { uint8_t *b = bytes(); for (size_t i = 0; i < n; i++) { d[i] = b[pos]; b[pos] = 0; if (++pos == R) run_f(); } }
    void begin_op(uint8_t flags, bool more) {
        if (more) return;
        uint8_t hdr[2] = {pos_begin, flags};
        pos_begin = (uint8_t)(pos + 1);
        absorb(hdr, 2);
        if ((flags & (FC | FK)) && pos != 0) run_f();
    }
    void meta_ad(const uint8_t *d, size_t n, bool more) { begin_op(FM | FA, more); absorb(d, n); }
    void ad(const uint8_t *d, size_t n, bool more) { begin_op(FA, more); absorb(d, n); }
    void prf(uint8_t *d, size_t n) { begin_op(FI | FA | FC, false); squeeze(d, n); }
    void key(const uint8_t *d, size_t n) { begin_op(FA | FC, false); overwrite(d, n); }
    void init(const char *proto) {
        memset(st, 0, sizeof st); pos = 0; pos_begin = 0;
        const uint8_t hdr[6] = {1, R + 2, 1, 0, 1, 96};
        memcpy(bytes(), hdr, 6); memcpy(bytes() + 6, "STROBEv1.0.2", 12);
        keccak_f(st);
        meta_ad((const uint8_t *)proto, strlen(proto), false);
    }
    void append_message(const char *label, const uint8_t *m, uint32_t n) {   // transcript.rs:69-74
        uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
        meta_ad((const uint8_t *)label, strlen(label), false); meta_ad(len, 4, true); ad(m, n, false);
    }
};

}  // namespace c25519_tr

// hrams: n x 64, sigs: n x 64 (s = bytes 32..63), zs out: n x 16
static void c25519_transcript_zs(const uint8_t *hrams, const uint8_t *sigs, uint64_t n, uint8_t *zs) {
    c25519_tr::strobe t;
    t.init("Merlin v1.0");                                                          // transcript.rs:54-58
    t.append_message("dom-sep", (const uint8_t *)"ed25519 batch verification", 26);  // batch.rs:168
    for (uint64_t i = 0; i < n; i++) t.append_message("hram", hrams + 64 * i, 64);   // batch.rs:195-197
    for (uint64_t i = 0; i < n; i++) t.append_message("sig.s", sigs + 64 * i + 32, 32);  // :199-201
    uint8_t zeros[32] = {0};                                                         // ZeroRng, batch.rs:49-76
    t.meta_ad((const uint8_t *)"rng", 3, false); t.key(zeros, 32);                   // transcript.rs:157-173
    const uint8_t len16[4] = {16, 0, 0, 0};
    for (uint64_t i = 0; i < n; i++) { t.meta_ad(len16, 4, false); t.prf(zs + 16 * i, 16); }  // transcript.rs:200-206
}
