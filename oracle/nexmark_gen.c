/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under flock_amd/ may include,
 * link or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it (as the checker, never as the thing measured).
 *
 * CPU restatement of Flock's NEXMark event generator, straight into Arrow-style
 * columns (no JSON detour).  Follows, field by field:
 *   flock/src/datasource/nexmark/event.rs:83-97   Event::new (kind by id % 50)
 *   flock/src/datasource/nexmark/event.rs:152-185 Person::new / next_id / last_id
 *   flock/src/datasource/nexmark/event.rs:247-311 Auction::new / next_id / last_id / next_length
 *   flock/src/datasource/nexmark/event.rs:354-371 Bid::new
 *   flock/src/datasource/nexmark/event.rs:34-55   gen_string / gen_price
 *   flock/src/datasource/nexmark/config.rs:121-157 defaults (proportions 1:3:46,
 *       first ids 1000, hot ratios 4/2/4, ratio_2 = 100, in-flight 100,
 *       active people 1000, id lead 10, 5 categories from 10, word lists)
 *   flock/src/datasource/nexmark/config.rs:247-252 event_timestamp
 *
 * Documented deviations (SURVEY.md section 8(d) allows them: oracle and GPU consume
 * the SAME generated columns, so the RNG bit stream is not a parity contract):
 *   D1. rand-0.8 SmallRng::seed_from_u64(id) is replaced by a counter-based
 *       generator: draw k of event `id` = mix64(mix64(seed ^ id*C1) + (k+1)*C2).
 *   D2. gen_price = round(10^(6u) * 100) is evaluated in integer fixed point
 *       (u = 24-bit draw, 257-entry 2^x table) so CPU and GPU agree bit for bit.
 *   D3. event_timestamp uses exact integer math base + n*1000/eps instead of
 *       f32 rounding (config.rs:250-251 loses integer precision above 2^24
 *       events), so every epoch holds exactly `eps` events.
 *   D4. One generator per stream (the source function forces threads = 1,
 *       flock-function/src/aws/nexmark/source.rs:44-48), `first_event_id`
 *       selects the slice of the global stream a shard owns.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#include "nexmark_exp2_table.h"

#define PERSON_PROPORTION 1
#define AUCTION_PROPORTION 3
#define BID_PROPORTION 46
#define PROPORTION_DENOMINATOR 50
#define FIRST_AUCTION_ID 1000
#define FIRST_PERSON_ID 1000
#define FIRST_CATEGORY_ID 10
#define NUM_CATEGORIES 5
#define HOT_SELLER_RATIO 4
#define HOT_AUCTION_RATIO 2
#define HOT_BIDDER_RATIO 4
#define HOT_RATIO_2 100
#define IN_FLIGHT_AUCTIONS 100
#define ACTIVE_PEOPLE 1000
#define AUCTION_ID_LEAD 10
#define PERSON_ID_LEAD 10

static const uint32_t EXP2_Q30[257] = NEXMARK_EXP2_TABLE_INIT;

static const char *US_STATES[6] = {"az", "ca", "id", "or", "wa", "wy"};
static const char *US_CITIES[10] = {"phoenix", "los angeles", "san francisco", "boise", "portland",
                                    "bend", "redmond", "seattle", "kent", "cheyenne"};
static const char *FIRST_NAMES[11] = {"peter", "paul", "luke", "john", "saul", "vicky",
                                      "kate", "julie", "sarah", "deiter", "walter"};
static const char *LAST_NAMES[9] = {"shultz", "abrams", "spencer", "white", "bartels",
                                    "walton", "smith", "jones", "noris"};

typedef struct {
    uint64_t seed;
    uint64_t first_event_id; /* config.rs first_event_id: global id of this stream's event 0 */
    uint64_t eps;            /* events per second (single generator)                         */
    uint64_t base_time;      /* ms since epoch, config.rs BASE_TIME                          */
} nexmark_stream_t;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
static inline uint64_t ev_base(uint64_t seed, uint64_t id) { return mix64(seed ^ (id * 0xD6E8FEB86659FD93ull)); }
static inline uint64_t draw(uint64_t base, uint32_t k) { return mix64(base + (uint64_t)(k + 1) * 0x9E3779B97F4A7C15ull); }
/* uniform integer in [0, n), n < 2^32 (multiply-shift on the high 32 bits) */
static inline uint64_t uni(uint64_t r, uint64_t n) { return ((r >> 32) * n) >> 32; }

/* D2: price = round(10^(6u) * 100), u = k / 2^24 */
static inline uint32_t price_from(uint64_t r) {
    const uint64_t F = 85605435163ull;             /* floor(6*log2(10) * 2^32) */
    uint64_t k = r >> 40;                          /* 24 bits */
    uint64_t t = (k * F) >> 24;                    /* exponent, 32 fractional bits */
    uint32_t ip = (uint32_t)(t >> 32);
    uint32_t fp = (uint32_t)t;
    uint32_t idx = fp >> 24, rem = (fp >> 8) & 0xFFFFu;
    uint64_t m = EXP2_Q30[idx] + ((((uint64_t)(EXP2_Q30[idx + 1] - EXP2_Q30[idx])) * rem) >> 16);
    return (uint32_t)((((100ull * m) << ip) + (1ull << 29)) >> 30);
}

static inline uint64_t ts_of(const nexmark_stream_t *s, uint64_t event_number) {
    return s->base_time + (event_number * 1000ull) / s->eps; /* D3 */
}

static inline uint64_t person_last_id(uint64_t id) { /* event.rs:177-184 with person_proportion = 1 */
    return id / PROPORTION_DENOMINATOR;
}
static inline uint64_t auction_last_id(uint64_t id) { /* event.rs:283-295 */
    uint64_t epoch = id / PROPORTION_DENOMINATOR, offset = id % PROPORTION_DENOMINATOR;
    if (offset < PERSON_PROPORTION) { epoch -= 1; offset = AUCTION_PROPORTION - 1; }
    else if (offset >= PERSON_PROPORTION + AUCTION_PROPORTION) offset = AUCTION_PROPORTION - 1;
    else offset -= PERSON_PROPORTION;
    return epoch * AUCTION_PROPORTION + offset;
}
static inline uint64_t person_next_id(uint64_t id, uint64_t r) { /* event.rs:171-175 */
    uint64_t people = person_last_id(id) + 1;
    uint64_t active = people < ACTIVE_PEOPLE ? people : ACTIVE_PEOPLE;
    return people - active + uni(r, active + PERSON_ID_LEAD);
}
static inline uint64_t auction_next_id(uint64_t id, uint64_t r) { /* event.rs:273-281 */
    uint64_t max_a = auction_last_id(id);
    uint64_t min_a = max_a < IN_FLIGHT_AUCTIONS ? 0 : max_a - IN_FLIGHT_AUCTIONS;
    return min_a + uni(r, max_a - min_a + 1 + AUCTION_ID_LEAD);
}

/* gen_string(max): len = U[3, max), each char ' ' w.p. 1/13 else 'a'+U[0,26); trimmed (event.rs:34-51) */
static size_t gen_string(uint64_t base, uint32_t k0, uint32_t max, char *out) {
    uint32_t len = 3 + (uint32_t)uni(draw(base, k0), max - 3);
    char tmp[128];
    for (uint32_t i = 0; i < len; ++i) {
        uint64_t r = draw(base, k0 + 1 + i);
        tmp[i] = uni(r, 13) == 0 ? ' ' : (char)('a' + (((r & 0xFFFFFFFFull) * 26) >> 32));
    }
    uint32_t b = 0, e = len;
    while (b < e && tmp[b] == ' ') ++b;
    while (e > b && tmp[e - 1] == ' ') --e;
    memcpy(out, tmp + b, e - b);
    return e - b;
}

/* ---- public API ------------------------------------------------------------------ */

/* Kind counts for event numbers [n0, n1) of a stream (event.rs:84-96). */
void oracle_nexmark_counts(uint64_t first_event_id, uint64_t n0, uint64_t n1,
                           uint64_t *n_person, uint64_t *n_auction, uint64_t *n_bid) {
    uint64_t p = 0, a = 0, b = 0;
    /* closed form over whole 50-blocks, loop over the ragged ends */
    for (uint64_t n = n0; n < n1;) {
        uint64_t id = first_event_id + n;
        if (id % PROPORTION_DENOMINATOR == 0 && n + PROPORTION_DENOMINATOR <= n1) {
            uint64_t blocks = (n1 - n) / PROPORTION_DENOMINATOR;
            p += blocks * PERSON_PROPORTION; a += blocks * AUCTION_PROPORTION; b += blocks * BID_PROPORTION;
            n += blocks * PROPORTION_DENOMINATOR;
            continue;
        }
        uint64_t rem = id % PROPORTION_DENOMINATOR;
        if (rem < PERSON_PROPORTION) ++p; else if (rem < PERSON_PROPORTION + AUCTION_PROPORTION) ++a; else ++b;
        ++n;
    }
    *n_person = p; *n_auction = a; *n_bid = b;
}

/* Bids of events [n0, n1): columns auction,bidder,price (Int32), b_date_time (Timestamp ms). Returns rows. */
uint64_t oracle_nexmark_gen_bids(const nexmark_stream_t *s, uint64_t n0, uint64_t n1,
                                 int32_t *auction, int32_t *bidder, int32_t *price, int64_t *date_time) {
    uint64_t row = 0;
    for (uint64_t n = n0; n < n1; ++n) {
        uint64_t id = s->first_event_id + n;
        if (id % PROPORTION_DENOMINATOR < PERSON_PROPORTION + AUCTION_PROPORTION) continue;
        uint64_t base = ev_base(s->seed, id);
        uint64_t a = uni(draw(base, 0), HOT_AUCTION_RATIO) > 0
                         ? (auction_last_id(id) / HOT_RATIO_2) * HOT_RATIO_2
                         : auction_next_id(id, draw(base, 1));
        uint64_t b = uni(draw(base, 2), HOT_BIDDER_RATIO) > 0
                         ? (person_last_id(id) / HOT_RATIO_2) * HOT_RATIO_2 + 1
                         : person_next_id(id, draw(base, 3));
        if (auction) auction[row] = (int32_t)(a + FIRST_AUCTION_ID);
        if (bidder) bidder[row] = (int32_t)(b + FIRST_PERSON_ID);
        if (price) price[row] = (int32_t)price_from(draw(base, 4));
        if (date_time) date_time[row] = (int64_t)ts_of(s, id);
        ++row;
    }
    return row;
}

/* Auctions of events [n0, n1).  String columns are optional (pass NULL offsets to skip).
 * item_off/desc_off have rows+1 entries; byte buffers sized rows*19 / rows*99 by the caller. */
uint64_t oracle_nexmark_gen_auctions(const nexmark_stream_t *s, uint64_t n0, uint64_t n1,
                                     int32_t *a_id, int32_t *initial_bid, int32_t *reserve,
                                     int64_t *a_date_time, int64_t *expires, int32_t *seller, int32_t *category,
                                     int32_t *item_off, char *item_bytes, int32_t *desc_off, char *desc_bytes) {
    uint64_t row = 0;
    int32_t ioff = 0, doff = 0;
    if (item_off) item_off[0] = 0;
    if (desc_off) desc_off[0] = 0;
    for (uint64_t n = n0; n < n1; ++n) {
        uint64_t id = s->first_event_id + n;
        uint64_t rem = id % PROPORTION_DENOMINATOR;
        if (rem < PERSON_PROPORTION || rem >= PERSON_PROPORTION + AUCTION_PROPORTION) continue;
        uint64_t base = ev_base(s->seed, id);
        uint32_t ib = price_from(draw(base, 0));
        uint64_t sel = uni(draw(base, 1), HOT_SELLER_RATIO) > 0
                           ? (person_last_id(id) / HOT_RATIO_2) * HOT_RATIO_2
                           : person_next_id(id, draw(base, 2));
        uint64_t time = ts_of(s, id);
        /* next_length, event.rs:297-310 */
        uint64_t events_for_auctions = (IN_FLIGHT_AUCTIONS * PROPORTION_DENOMINATOR) / AUCTION_PROPORTION;
        uint64_t horizon = ts_of(s, id + events_for_auctions) - time;
        uint64_t span = horizon * 2 > 1 ? horizon * 2 : 1;
        if (a_id) a_id[row] = (int32_t)(auction_last_id(id) + FIRST_AUCTION_ID);
        if (initial_bid) initial_bid[row] = (int32_t)ib;
        if (reserve) reserve[row] = (int32_t)(ib + price_from(draw(base, 4)));
        if (a_date_time) a_date_time[row] = (int64_t)time;
        if (expires) expires[row] = (int64_t)(time + 1 + uni(draw(base, 5), span));
        if (seller) seller[row] = (int32_t)(sel + FIRST_PERSON_ID);
        if (category) category[row] = (int32_t)(FIRST_CATEGORY_ID + uni(draw(base, 6), NUM_CATEGORIES));
        if (item_off) { ioff += (int32_t)gen_string(base, 16, 20, item_bytes + ioff); item_off[row + 1] = ioff; }
        if (desc_off) { doff += (int32_t)gen_string(base, 40, 100, desc_bytes + doff); desc_off[row + 1] = doff; }
        ++row;
    }
    return row;
}

/* Persons of events [n0, n1).  All Utf8 columns = int32 offsets (rows+1) + bytes.
 * Caller-sized byte buffers: name rows*14, email rows*15, credit rows*19, city rows*13, state rows*2.
 * email/credit may be NULL (filler never read by q1/q2/q3/q5/q8). */
uint64_t oracle_nexmark_gen_persons(const nexmark_stream_t *s, uint64_t n0, uint64_t n1,
                                    int32_t *p_id, int64_t *p_date_time,
                                    int32_t *name_off, char *name_bytes,
                                    int32_t *email_off, char *email_bytes,
                                    int32_t *cc_off, char *cc_bytes,
                                    int32_t *city_off, char *city_bytes,
                                    int32_t *state_off, char *state_bytes) {
    uint64_t row = 0;
    int32_t no = 0, eo = 0, co = 0, cio = 0, so = 0;
    if (name_off) name_off[0] = 0;
    if (email_off) email_off[0] = 0;
    if (cc_off) cc_off[0] = 0;
    if (city_off) city_off[0] = 0;
    if (state_off) state_off[0] = 0;
    for (uint64_t n = n0; n < n1; ++n) {
        uint64_t id = s->first_event_id + n;
        if (id % PROPORTION_DENOMINATOR >= PERSON_PROPORTION) continue;
        uint64_t base = ev_base(s->seed, id);
        if (p_id) p_id[row] = (int32_t)(person_last_id(id) + FIRST_PERSON_ID);
        if (p_date_time) p_date_time[row] = (int64_t)ts_of(s, id);
        if (name_off) {
            const char *f = FIRST_NAMES[uni(draw(base, 0), 11)], *l = LAST_NAMES[uni(draw(base, 1), 9)];
            size_t fl = strlen(f), ll = strlen(l);
            memcpy(name_bytes + no, f, fl); name_bytes[no + fl] = ' '; memcpy(name_bytes + no + fl + 1, l, ll);
            no += (int32_t)(fl + 1 + ll); name_off[row + 1] = no;
        }
        if (city_off) {
            const char *c = US_CITIES[uni(draw(base, 2), 10)]; size_t cl = strlen(c);
            memcpy(city_bytes + cio, c, cl); cio += (int32_t)cl; city_off[row + 1] = cio;
        }
        if (state_off) {
            const char *st = US_STATES[uni(draw(base, 3), 6)];
            memcpy(state_bytes + so, st, 2); so += 2; state_off[row + 1] = so;
        }
        if (email_off) {
            eo += (int32_t)gen_string(base, 8, 7, email_bytes + eo); email_bytes[eo++] = '@';
            eo += (int32_t)gen_string(base, 20, 5, email_bytes + eo); memcpy(email_bytes + eo, ".com", 4); eo += 4;
            email_off[row + 1] = eo;
        }
        if (cc_off) {
            for (int g = 0; g < 4; ++g) {
                uint32_t v = (uint32_t)uni(draw(base, 30 + g), 10000);
                if (g) cc_bytes[co++] = ' ';
                cc_bytes[co++] = (char)('0' + v / 1000); cc_bytes[co++] = (char)('0' + (v / 100) % 10);
                cc_bytes[co++] = (char)('0' + (v / 10) % 10); cc_bytes[co++] = (char)('0' + v % 10);
            }
            cc_off[row + 1] = co;
        }
        ++row;
    }
    return row;
}
