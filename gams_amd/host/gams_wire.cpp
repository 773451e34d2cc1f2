// gams_wire.cpp -- the byte formats either side of the hot path: what the reference
// reads from and writes to Redis for this path.  Host layer only; no server, no store.
//
//   bincode 1.3.3 (Cargo.toml:37, default options: little endian, fixed-width integers, u64 lengths)
//     BTreeMap<String, Ctg>  under bundle:ctg:{chr}   src/libs/redis.rs:216-233, src/libs/data.rs:5-14
//     Lapper<u32, String>    under idx:ctg:{chr} / idx:rg:{ctg}   src/libs/redis.rs:236-324
//                            (rust-lapper 1.1.0 with_serde, Cargo.toml:38)
//   RESP2 (redis 0.25.4, Cargo.toml:24): commands as arrays of bulk strings, the five reply kinds;
//     GET / SET / INCR (redis.rs:98-161), the SET pipeline (redis.rs:385-407), EVAL of the SCAN
//     scripts (redis.rs:332-383).
//
// PARITY UNPINNED: bincode, rust-lapper and the redis crate are not part of /root/reference and no
// fixture of the reference holds these bytes; the layouts below restate the crates' published
// formats (serde derive order of the structs in data.rs / rust-lapper's struct Lapper).  Tests
// compare against vectors assembled by hand from those rules.
#include <algorithm>
#include <cstring>

#include "gams_host.hpp"

namespace gams {
namespace wire {

namespace {

void put_u64(std::string &o, uint64_t v) {
    for (int i = 0; i < 8; ++i) o.push_back((char)((v >> (8 * i)) & 0xff));
}
void put_u32(std::string &o, uint32_t v) {
    for (int i = 0; i < 4; ++i) o.push_back((char)((v >> (8 * i)) & 0xff));
}
void put_str(std::string &o, const std::string &s) {   // String: u64 length + UTF-8 bytes
    put_u64(o, s.size());
    o += s;
}

// std::str::from_utf8: shortest forms only, no surrogates, nothing above U+10FFFF
bool valid_utf8(const uint8_t *b, size_t n) {
    for (size_t i = 0; i < n;) {
        const uint8_t c = b[i];
        size_t k;
        uint32_t lo;
        if (c < 0x80) { ++i; continue; }
        if (c >= 0xC2 && c <= 0xDF) { k = 1; lo = 0x80; }
        else if (c >= 0xE0 && c <= 0xEF) { k = 2; lo = 0x800; }
        else if (c >= 0xF0 && c <= 0xF4) { k = 3; lo = 0x10000; }
        else return false;
        if (n - i <= k) return false;
        uint32_t cp = c & (0x3Fu >> k);
        for (size_t j = 1; j <= k; ++j) {
            if ((b[i + j] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (b[i + j] & 0x3Fu);
        }
        if (cp < lo || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += k + 1;
    }
    return true;
}

struct Reader {
    const uint8_t *p;
    size_t n, at = 0;
    void need(size_t k) const {
        if (n - at < k) throw Error(GAMS_EINVAL, "bincode: input ends inside a value (offset " + std::to_string(at) + ")");
    }
    uint64_t u64() {
        need(8);
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v |= (uint64_t)p[at + i] << (8 * i);
        at += 8;
        return v;
    }
    uint32_t u32() {
        need(4);
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i) v |= (uint32_t)p[at + i] << (8 * i);
        at += 4;
        return v;
    }
    uint8_t u8() {
        need(1);
        return p[at++];
    }
    std::string str() {
        const uint64_t len = u64();
        if (len > n - at) throw Error(GAMS_EINVAL, "bincode: string length " + std::to_string(len) + " beyond the input");
        std::string s(reinterpret_cast<const char *>(p + at), (size_t)len);
        // a Rust String: bincode's deserializer refuses bytes that are not UTF-8 ("invalid utf-8 encoding")
        if (!valid_utf8(p + at, (size_t)len))
            throw Error(GAMS_EINVAL, "bincode: string at offset " + std::to_string(at) + " is not UTF-8");
        at += (size_t)len;
        return s;
    }
    uint64_t count(size_t min_item_bytes) {   // a collection length that the remaining input can hold
        const uint64_t c = u64();
        if (min_item_bytes && c > (n - at) / min_item_bytes)
            throw Error(GAMS_EINVAL, "bincode: collection of " + std::to_string(c) + " items beyond the input");
        return c;
    }
};

}  // namespace

// ---- bundle:ctg:{chr} -------------------------------------------------------------------------
// BTreeMap<String, Ctg>: u64 entry count, then (key, value) in key order (Rust String order =
// byte-wise); Ctg = id, range, chr_id : String; chr_start, chr_end : i32; chr_strand : String; length : i32.
std::string bincode_ctg_bundle(const std::vector<Ctg> &ctgs) {
    std::vector<const Ctg *> order;
    for (const Ctg &c : ctgs) order.push_back(&c);
    std::stable_sort(order.begin(), order.end(), [](const Ctg *a, const Ctg *b) { return a->id < b->id; });
    // a map holds one value per key: the last insert wins (BTreeMap::insert)
    std::vector<const Ctg *> uniq;
    for (const Ctg *c : order) {
        if (!uniq.empty() && uniq.back()->id == c->id)
            uniq.back() = c;
        else
            uniq.push_back(c);
    }
    std::string o;
    put_u64(o, uniq.size());
    for (const Ctg *c : uniq) {
        put_str(o, c->id);           // key
        put_str(o, c->id);
        put_str(o, c->range);
        put_str(o, c->chr_id);
        put_u32(o, (uint32_t)c->chr_start);
        put_u32(o, (uint32_t)c->chr_end);
        put_str(o, c->chr_strand);
        put_u32(o, (uint32_t)c->length);
    }
    return o;
}

std::vector<Ctg> bincode_ctg_bundle_decode(const uint8_t *bytes, size_t n) {
    Reader r{bytes, n};
    const uint64_t cnt = r.count(8 + 8 * 4 + 12);
    // bincode hands the entries to BTreeMap one by one: any order is accepted, a repeated key keeps the
    // LAST value, and iteration (what get_bundle_ctg's callers see, redis.rs:216-233) is in key order.  The
    // map is keyed by the stored KEY; the reference writes key == id, but nothing in the format says so.
    std::map<std::string, Ctg> by_key;
    for (uint64_t i = 0; i < cnt; ++i) {
        std::string key = r.str();
        Ctg c;
        c.id = r.str();
        c.range = r.str();
        c.chr_id = r.str();
        c.chr_start = (int32_t)r.u32();
        c.chr_end = (int32_t)r.u32();
        c.chr_strand = r.str();
        c.length = (int32_t)r.u32();
        by_key.insert_or_assign(std::move(key), std::move(c));
    }
    std::vector<Ctg> out;
    out.reserve(by_key.size());
    for (auto &kv : by_key) out.push_back(std::move(kv.second));
    if (r.at != n) throw Error(GAMS_EINVAL, "bincode: " + std::to_string(n - r.at) + " trailing bytes after the map");
    return out;
}

// ---- idx:ctg:{chr} / idx:rg:{ctg} ---------------------------------------------------------------
// Lapper::new(ivs): intervals.sort() (by start, then stop; stable), starts and stops sorted on their
// own, max_len = max(stop - start).  Serialised in struct order:
//   intervals: Vec<Interval{start u32, stop u32, val String}>, starts: Vec<u32>, stops: Vec<u32>,
//   max_len: u32, cov: Option<u32> (None = one 0 byte), overlaps_merged: bool (one byte)
std::string bincode_lapper(const std::vector<LapperIv> &ivs_in) {
    std::vector<LapperIv> ivs = ivs_in;
    std::stable_sort(ivs.begin(), ivs.end(), [](const LapperIv &a, const LapperIv &b) {
        return a.start != b.start ? a.start < b.start : a.stop < b.stop;
    });
    std::vector<uint32_t> starts, stops;
    uint32_t max_len = 0;
    for (const LapperIv &v : ivs) {
        starts.push_back(v.start);
        stops.push_back(v.stop);
        if (v.stop > v.start) max_len = std::max(max_len, v.stop - v.start);   // checked_sub(..).unwrap_or(0)
    }
    std::sort(starts.begin(), starts.end());
    std::sort(stops.begin(), stops.end());
    std::string o;
    put_u64(o, ivs.size());
    for (const LapperIv &v : ivs) {
        put_u32(o, v.start);
        put_u32(o, v.stop);
        put_str(o, v.val);
    }
    put_u64(o, starts.size());
    for (uint32_t x : starts) put_u32(o, x);
    put_u64(o, stops.size());
    for (uint32_t x : stops) put_u32(o, x);
    put_u32(o, max_len);
    o.push_back('\0');   // cov: None
    o.push_back('\0');   // overlaps_merged: false
    return o;
}

LapperBlob bincode_lapper_decode(const uint8_t *bytes, size_t n) {
    Reader r{bytes, n};
    LapperBlob b;
    const uint64_t cnt = r.count(16);
    b.intervals.reserve((size_t)cnt);
    for (uint64_t i = 0; i < cnt; ++i) {
        LapperIv v;
        v.start = r.u32();
        v.stop = r.u32();
        v.val = r.str();
        b.intervals.push_back(std::move(v));
    }
    const uint64_t ns = r.count(4);
    for (uint64_t i = 0; i < ns; ++i) b.starts.push_back(r.u32());
    const uint64_t nt = r.count(4);
    for (uint64_t i = 0; i < nt; ++i) b.stops.push_back(r.u32());
    b.max_len = r.u32();
    const uint8_t tag = r.u8();
    if (tag > 1) throw Error(GAMS_EINVAL, "bincode: Option tag " + std::to_string(tag));
    b.has_cov = tag == 1;
    if (b.has_cov) b.cov = r.u32();
    const uint8_t flag = r.u8();
    if (flag > 1) throw Error(GAMS_EINVAL, "bincode: bool byte " + std::to_string(flag));
    b.overlaps_merged = flag == 1;
    if (r.at != n) throw Error(GAMS_EINVAL, "bincode: " + std::to_string(n - r.at) + " trailing bytes after the Lapper");
    if (ns != cnt || nt != cnt) throw Error(GAMS_EINVAL, "bincode: Lapper starts/stops do not match its intervals");
    return b;
}

// The device index over a set of decoded idx: blobs (one group per blob, in the order given): what
// get_idx_rg / get_idx_ctg (redis.rs:260-273, 305-324) hand to count_rg / find_one_idx.
gams_index_t *index_from_lappers(gams_gpu_t *h, const std::vector<LapperBlob> &blobs) {
    std::vector<uint64_t> off(blobs.size() + 1, 0);
    for (size_t g = 0; g < blobs.size(); ++g) off[g + 1] = off[g] + blobs[g].intervals.size();
    std::vector<uint32_t> starts, stops;
    starts.reserve((size_t)off.back());
    stops.reserve((size_t)off.back());
    for (const LapperBlob &b : blobs)
        for (const LapperIv &v : b.intervals) {
            starts.push_back(v.start);
            stops.push_back(v.stop);
        }
    gams_index_t *ix = nullptr;
    const int rc = gams_index_create(h, (uint32_t)blobs.size(), off.data(), starts.data(), stops.data(), &ix);
    if (rc != GAMS_OK) throw Error(rc, gams_gpu_last_error(h));
    return ix;
}

// ---- RESP2 ---------------------------------------------------------------------------------------
std::string resp_command(const std::vector<std::string> &args) {
    std::string o = "*" + std::to_string(args.size()) + "\r\n";
    for (const std::string &a : args) {
        o += "$" + std::to_string(a.size()) + "\r\n";
        o += a;
        o += "\r\n";
    }
    return o;
}

// redis::pipe().set(k, v).ignore()... : the commands back to back, one +OK per command comes back
std::string resp_pipeline_set(const std::vector<std::pair<std::string, std::string>> &kv) {
    std::string o;
    for (const auto &p : kv) o += resp_command({"SET", p.first, p.second});
    return o;
}

// Script::invoke of the redis crate tries EVALSHA and falls back to EVAL on NOSCRIPT; this is the
// EVAL form: EVAL script numkeys [key ...] [arg ...]
std::string resp_eval(const std::string &script, const std::vector<std::string> &keys,
                      const std::vector<std::string> &argv) {
    std::vector<std::string> a{"EVAL", script, std::to_string(keys.size())};
    a.insert(a.end(), keys.begin(), keys.end());
    a.insert(a.end(), argv.begin(), argv.end());
    return resp_command(a);
}

const char *scan_values_script() {   // src/libs/redis.rs:367-383, byte for byte
    return "\nlocal cursor = \"0\";\nlocal list = {};\nrepeat\n"
           "    local result = redis.call('SCAN', cursor, 'MATCH', ARGV[1], 'COUNT', ARGV[2])\n"
           "    cursor = result[1];\n    for _, key in ipairs(result[2]) do\n"
           "        list[#list+1] = redis.call('GET', key)\n    end\nuntil cursor == \"0\";\nreturn list;\n";
}

namespace {
// one line up to CRLF starting at `at`; npos when the line is not complete yet
size_t line_end(const char *p, size_t n, size_t at) {
    for (size_t i = at; i + 1 < n; ++i)
        if (p[i] == '\r' && p[i + 1] == '\n') return i;
    return std::string::npos;
}
bool parse_int(const char *p, size_t a, size_t b, int64_t &v) {
    if (a >= b) return false;
    bool neg = false;
    size_t i = a;
    if (p[i] == '-') {
        neg = true;
        ++i;
    } else if (p[i] == '+') {
        ++i;
    }
    if (i >= b) return false;
    uint64_t acc = 0;
    for (; i < b; ++i) {
        if (p[i] < '0' || p[i] > '9') return false;
        if (acc > (UINT64_MAX - 9) / 10) return false;
        acc = acc * 10 + (uint64_t)(p[i] - '0');
    }
    if (acc > (uint64_t)INT64_MAX + (neg ? 1u : 0u)) return false;
    v = neg ? (int64_t)(0 - acc) : (int64_t)acc;
    return true;
}
size_t parse_at(const char *p, size_t n, size_t at, RespValue &out, int depth) {
    if (depth > 32) throw Error(GAMS_EINVAL, "RESP: arrays nested deeper than 32");
    if (at >= n) return 0;
    const char kind = p[at];
    const size_t eol = line_end(p, n, at + 1);
    if (eol == std::string::npos) return 0;                    // header line incomplete
    out = RespValue();
    switch (kind) {
        case '+':
        case '-':
            out.type = kind == '+' ? RespValue::Simple : RespValue::Error;
            out.str.assign(p + at + 1, eol - at - 1);
            return eol + 2;
        case ':':
            out.type = RespValue::Integer;
            if (!parse_int(p, at + 1, eol, out.integer)) throw Error(GAMS_EINVAL, "RESP: bad integer");
            return eol + 2;
        case '$': {
            int64_t len;
            if (!parse_int(p, at + 1, eol, len) || len < -1) throw Error(GAMS_EINVAL, "RESP: bad bulk length");
            if (len == -1) {
                out.type = RespValue::Null;
                return eol + 2;
            }
            const size_t body = eol + 2;
            if (n - body < (uint64_t)len + 2) return 0;            // body incomplete
            if (p[body + len] != '\r' || p[body + len + 1] != '\n') throw Error(GAMS_EINVAL, "RESP: bulk string not closed by CRLF");
            out.type = RespValue::Bulk;
            out.str.assign(p + body, (size_t)len);
            return body + (size_t)len + 2;
        }
        case '*': {
            int64_t cnt;
            if (!parse_int(p, at + 1, eol, cnt) || cnt < -1) throw Error(GAMS_EINVAL, "RESP: bad array length");
            if (cnt == -1) {
                out.type = RespValue::Null;
                return eol + 2;
            }
            out.type = RespValue::Array;
            size_t cur = eol + 2;
            for (int64_t i = 0; i < cnt; ++i) {
                RespValue item;
                const size_t next = parse_at(p, n, cur, item, depth + 1);
                if (next == 0) return 0;
                out.array.push_back(std::move(item));
                cur = next;
            }
            return cur;
        }
        default:
            throw Error(GAMS_EINVAL, std::string("RESP: unknown type byte '") + kind + "'");
    }
}
}  // namespace

size_t resp_parse(const char *bytes, size_t n, RespValue &out) { return parse_at(bytes, n, 0, out, 0); }

// flat text form for tests and logs: one token per line, arrays as "*N"
std::string resp_dump(const RespValue &v) {
    switch (v.type) {
        case RespValue::Simple: return "+" + v.str + "\n";
        case RespValue::Error: return "-" + v.str + "\n";
        case RespValue::Integer: return ":" + std::to_string(v.integer) + "\n";
        case RespValue::Null: return "_\n";
        case RespValue::Bulk: return "$" + std::to_string(v.str.size()) + " " + v.str + "\n";
        case RespValue::Array: {
            std::string o = "*" + std::to_string(v.array.size()) + "\n";
            for (const RespValue &x : v.array) o += resp_dump(x);
            return o;
        }
    }
    return "";
}

}  // namespace wire
}  // namespace gams
