/* qname_collision.c -- finds two DIFFERENT read names with the same rsqc_qname_hash (FNV-1a 64 + fmix64; the finaliser is a
 * bijection, so a collision of the hash is a collision of FNV-1a).  Parallel collision search with distinguished points
 * (van Oorschot & Wiener): walk x -> fnv1a(name_of(x)) until the value has its low DP_BITS bits clear, remember
 * (distinguished point -> start of the trail); two trails that end in the same point merged somewhere: re-walk both to the
 * step where they join, which yields the two names.  About sqrt(pi/2 * 2^64) = 5.4e9 hash evaluations: a few minutes on
 * 8 threads.  The pair it printed is committed as tests/golden/qname_hash_collision.json (a fixture: data, found once).
 *   gcc -O2 -pthread tools/qname_collision.c -o /tmp/qname_collision && /tmp/qname_collision
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define DP_BITS 22
#define NAME_LEN 14
static const char ALPHA[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789:_";   /* 64 symbols */

static void name_of(uint64_t x, char *out) {            /* "r" + 11 symbols (66 bits >= 64: injective) + ":1" */
    out[0] = 'r';
    for (int i = 0; i < 11; ++i) { out[1 + i] = ALPHA[x & 63]; x >>= 6; }
    out[12] = '/'; out[13] = '1';
}
static uint64_t fnv(const char *s, int n) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (int i = 0; i < n; ++i) { h ^= (uint8_t)s[i]; h *= 0x100000001B3ull; }
    return h;
}
static uint64_t step(uint64_t x) { char b[NAME_LEN]; name_of(x, b); return fnv(b, NAME_LEN); }
static uint64_t full_hash(const char *s, int n) {
    uint64_t h = fnv(s, n);
    h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33;
    return h;
}

#define TAB_BITS 24
typedef struct { uint64_t dp, start, len; } Ent;
static Ent *tab;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static volatile int done = 0;
static uint64_t res_a, res_b;

static int resolve(uint64_t s1, uint64_t l1, uint64_t s2, uint64_t l2) {
    while (l1 > l2) { s1 = step(s1); --l1; }
    while (l2 > l1) { s2 = step(s2); --l2; }
    if (s1 == s2) return 0;                              /* one trail is a suffix of the other (same start): no collision */
    for (uint64_t i = 0; i < l1; ++i) {
        const uint64_t n1 = step(s1), n2 = step(s2);
        if (n1 == n2) { res_a = s1; res_b = s2; return 1; }
        s1 = n1; s2 = n2;
    }
    return 0;
}
static void *worker(void *arg) {
    uint64_t seed = 0x9E3779B97F4A7C15ull * (uint64_t)(1 + (intptr_t)arg);
    while (!done) {
        seed = seed * 6364136223846793005ull + 1442695040888963407ull;
        uint64_t start = seed, x = start, len = 0;
        while (len < (20ull << DP_BITS)) {
            x = step(x); ++len;
            if ((x & ((1ull << DP_BITS) - 1)) == 0) break;
        }
        if ((x & ((1ull << DP_BITS) - 1)) != 0) continue;            /* abandoned (a cycle without a distinguished point) */
        pthread_mutex_lock(&mu);
        uint64_t slot = (x >> DP_BITS) & ((1ull << TAB_BITS) - 1);
        for (;; slot = (slot + 1) & ((1ull << TAB_BITS) - 1)) {
            if (tab[slot].len == 0) { tab[slot].dp = x; tab[slot].start = start; tab[slot].len = len; break; }
            if (tab[slot].dp == x) {
                const Ent e = tab[slot];
                pthread_mutex_unlock(&mu);
                if (!done && resolve(e.start, e.len, start, len)) { done = 1; return NULL; }
                pthread_mutex_lock(&mu);
                break;
            }
        }
        pthread_mutex_unlock(&mu);
    }
    return NULL;
}
int main(int argc, char **argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 8;
    tab = calloc((size_t)1 << TAB_BITS, sizeof(Ent));
    pthread_t th[64];
    for (int i = 0; i < nt; ++i) pthread_create(&th[i], NULL, worker, (void *)(intptr_t)i);
    for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
    char a[NAME_LEN + 1], b[NAME_LEN + 1];
    name_of(res_a, a); name_of(res_b, b); a[NAME_LEN] = b[NAME_LEN] = 0;
    printf("{\"a\": \"%s\", \"b\": \"%s\", \"hash_a\": \"%016llx\", \"hash_b\": \"%016llx\"}\n", a, b,
           (unsigned long long)full_hash(a, NAME_LEN), (unsigned long long)full_hash(b, NAME_LEN));
    return strcmp(a, b) != 0 && full_hash(a, NAME_LEN) == full_hash(b, NAME_LEN) ? 0 : 1;
}
