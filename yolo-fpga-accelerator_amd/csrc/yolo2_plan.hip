// yolo2_plan.hip -- planning data of libyolo2_hip.so that is not kernel code: the per-context option set (Y2Options), the
// committed plan table (config/plan_gfx950.txt), the weight-side plan cache (<weights>.y2plan) and the
// K-split scratch rule.  Host code only; everything here runs without a GPU (tests/test_host_logic.py drives it through the
// exported checks).
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <unistd.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>

#include "y2_internal.hpp"

// ---------------------------------------------------------------------------- options

namespace {
struct FlagOpt { const char *name; bool Y2Options::*m; };
struct IntOpt { const char *name; int Y2Options::*m; int lo, hi; };
struct StrOpt { const char *name; std::string Y2Options::*m; };

const FlagOpt kFlags[] = {
    {"no_lanes", &Y2Options::no_lanes}, {"verbose", &Y2Options::verbose}, {"no_plan_cache", &Y2Options::no_plan_cache},
    {"no_poolfuse", &Y2Options::no_poolfuse}, {"no_hiacc", &Y2Options::no_hiacc}, {"no_ks", &Y2Options::no_ks},
    {"no_w16", &Y2Options::no_w16}, {"no_grp", &Y2Options::no_grp}, {"grp16", &Y2Options::grp16}, {"no_xcd_remap", &Y2Options::no_xcd_remap},
    {"splitk_no_pack", &Y2Options::splitk_no_pack}, {"force_w16", &Y2Options::force_w16}, {"force_hiacc", &Y2Options::force_hiacc},
    {"f16_no_lanes", &Y2Options::f16_no_lanes}, {"f16_no_mfma0", &Y2Options::f16_no_mfma0}, {"f16_no_glds", &Y2Options::f16_no_glds},
    {"f16_no_poolfuse", &Y2Options::f16_no_poolfuse}, {"f16_no_halo", &Y2Options::f16_no_halo}, {"f16_no_persist", &Y2Options::f16_no_persist},
    {"f16_persist_all", &Y2Options::f16_persist_all}, {"f16_ring_all", &Y2Options::f16_ring_all}, {"f16_no_ring", &Y2Options::f16_no_ring},
    {"f16_no_c32", &Y2Options::f16_no_c32}, {"f16_m16", &Y2Options::f16_m16}, {"f16_w8", &Y2Options::f16_w8}, {"f16_no_wide", &Y2Options::f16_no_wide},
    {"f16_no_fuse1x1", &Y2Options::f16_no_fuse1x1}, {"f16_no_rw", &Y2Options::f16_no_rw}, {"f16_no_rwb", &Y2Options::f16_no_rwb}, {"f16_no_rwc", &Y2Options::f16_no_rwc}, {"f16_ring256", &Y2Options::f16_ring256}, {"f16_ring_sq", &Y2Options::f16_ring_sq},
};
const IntOpt kInts[] = {
    {"autotune", &Y2Options::autotune, -1, 1}, {"lanes", &Y2Options::lanes, 0, 8}, {"lane_priority", &Y2Options::lane_priority, 0, 1},
    {"splitk", &Y2Options::splitk, -1, 1}, {"poolfuse", &Y2Options::poolfuse, -1, 1}, {"f16_lanes", &Y2Options::f16_lanes, 1, 8},
    {"stamp_layer", &Y2Options::stamp_layer, -1, 31}, {"f16_skip", &Y2Options::f16_skip, 0, 63}, {"force_path", &Y2Options::force_path, -1, 4}, {"force_p", &Y2Options::force_p, 0, 8},
    {"force_ks", &Y2Options::force_ks, 0, 16}, {"f32_p", &Y2Options::f32_p, 0, 4},
};
const StrOpt kStrs[] = {
    {"plan_file", &Y2Options::plan_file}, {"plan_write", &Y2Options::plan_write}, {"lane_split", &Y2Options::lane_split},
};
}  // namespace

int Y2Options::set(const char *name, const char *value)
{
    if (!name) return -1;
    const Y2Options dflt;
    const bool clear = !value || !value[0];
    for (const FlagOpt &f : kFlags)
        if (!strcmp(name, f.name)) {   // a flag is on for any value except "0"
            this->*f.m = clear ? dflt.*f.m : strcmp(value, "0") != 0;
            return 0;
        }
    for (const IntOpt &o : kInts)
        if (!strcmp(name, o.name)) {
            if (clear) { this->*o.m = dflt.*o.m; return 0; }
            char *end = nullptr;
            const long v = strtol(value, &end, 10);
            if (end == value || *end || v < o.lo || v > o.hi) return -1;
            this->*o.m = (int)v;
            return 0;
        }
    for (const StrOpt &o : kStrs)
        if (!strcmp(name, o.name)) {
            this->*o.m = clear ? std::string() : std::string(value);
            return 0;
        }
    return -1;
}

Y2Options Y2Options::from_env()
{
    Y2Options o;
    auto env_name = [](const char *name) {
        std::string e = "YOLO2_";
        for (const char *p = name; *p; ++p) e += (char)toupper((unsigned char)*p);
        return e;
    };
    auto take = [&](const char *name) {
        const char *v = getenv(env_name(name).c_str());
        if (!v) return;
        // (historical spelling: a flag variable that is present but empty counts as set)
        if (o.set(name, v[0] ? v : "1") != 0)
            fprintf(stderr, "[yolo2_hip] ignoring %s=%s (out of range)\n", env_name(name).c_str(), v);
    };
    for (const FlagOpt &f : kFlags) take(f.name);
    for (const IntOpt &i : kInts) take(i.name);
    for (const StrOpt &s : kStrs) take(s.name);
    return o;
}

std::string Y2Options::describe() const
{
    const Y2Options d;
    std::string out;
    auto add = [&](const std::string &kv) { out += (out.empty() ? "" : " ") + kv; };
    for (const FlagOpt &f : kFlags)
        if (this->*f.m != d.*f.m) add(std::string(f.name) + "=" + (this->*f.m ? "1" : "0"));
    for (const IntOpt &i : kInts)
        if (this->*i.m != d.*i.m) add(std::string(i.name) + "=" + std::to_string(this->*i.m));
    for (const StrOpt &s : kStrs)
        if (this->*s.m != d.*s.m) add(std::string(s.name) + "=" + this->*s.m);
    return out;
}

static Y2Options &process_options_mut()
{
    static Y2Options o = Y2Options::from_env();
    return o;
}
const Y2Options &y2_process_options() { return process_options_mut(); }

extern "C" int yolo2_hip_set_option(yolo2_hip_ctx *c, const char *name, const char *value)
{
    if (!name) return fail(YOLO2_ERROR, "null argument");
    if (!c) {   // the context-less driver tier (yolo2_execute_conv_layer ...): the process-wide set; not thread-safe against running layer calls
        if (process_options_mut().set(name, value) != 0) return fail(YOLO2_ERROR, "unknown option or value out of range: %s=%s", name, value ? value : "");
        return YOLO2_SUCCESS;
    }
    if (c->opt.set(name, value) != 0) return fail(YOLO2_ERROR, "unknown option or value out of range: %s=%s", name, value ? value : "");
    for (yolo2_hip_ctx *l : c->lanes) l->opt = c->opt;
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_options_string(yolo2_hip_ctx *c, char *buf, int cap)
{
    if (!c || !buf || cap <= 0) return fail(YOLO2_ERROR, "null argument");
    snprintf(buf, (size_t)cap, "%s", c->opt.describe().c_str());
    return YOLO2_SUCCESS;
}

// ---------------------------------------------------------------------------- the K-split scratch rule

size_t y2_ks_bytes(int splits, int cg_out, int npix) { return (size_t)splits * (size_t)cg_out * (size_t)npix * 24; }

bool y2_ks_fits(int splits, int cg_out, int npix, size_t cap_bytes)
{
    if (splits <= 0 || cg_out <= 0 || npix <= 0 || cap_bytes == 0) return false;
    return y2_ks_bytes(splits, cg_out, npix) <= cap_bytes;
}

// k_conv_i16_ks stores, for each of its `splits` sub-chains, one 24-byte clamp-affine triple per output item (4 channels) and pixel
// into the context's scratch; k_ks_finalize reads them back.  Round 3's GPU fault (a batch-64 run with the variant forced; DESIGN.md
// 4.1) was this rule violated twice over: a scratch sized for two layer shapes, and a `ks` plan field alive in a context that had
// never allocated the scratch (cap = 0).  The rule on plain numbers, no GPU needed: YOLO2_SUCCESS iff the plan may be launched.
extern "C" int yolo2_hip_i16_plan_check(int splits, int cg_out, int npix, size_t cap_bytes)
{
    if (splits != 2 && splits != 4 && splits != 8 && splits != 16) return fail(YOLO2_ERROR, "K-split: %d splits (2, 4, 8 or 16)", splits);
    if (cg_out <= 0 || npix <= 0) return fail(YOLO2_ERROR, "K-split: empty output (%d items x %d pixels)", cg_out, npix);
    if (cap_bytes == 0) return fail(YOLO2_ERROR, "K-split: this context holds no triple scratch (batch > 4 or scratch not allocated): the plan is refused");
    if (!y2_ks_fits(splits, cg_out, npix, cap_bytes))
        return fail(YOLO2_ERROR, "K-split: %d splits x %d items x %d pixels x 24 B = %zu bytes of triples do not fit the %zu-byte scratch", splits, cg_out,
                    npix, y2_ks_bytes(splits, cg_out, npix), cap_bytes);
    return YOLO2_SUCCESS;
}

// ---------------------------------------------------------------------------- plan lines and the committed table

bool y2_plan_line_parse(const char *line, Y2PlanKey *key, Y2PlanLine *out)
{
    int B, L, S;
    Y2PlanLine pl;
    pl.hiacc = pl.ks = 0;
    if (line[0] == '#' || sscanf(line, "%d %d %d %d %d %d %d %d %d %d %d %d", &B, &L, &S, &pl.path, &pl.P, &pl.pad, &pl.splitk, &pl.pp, &pl.w16, &pl.fuse,
                                 &pl.hiacc, &pl.ks) < 10)
        return false;
    auto one_of = [](int v, std::initializer_list<int> ok) { for (int o : ok) if (v == o) return true; return false; };
    if (B <= 0 || B > 4096 || L < 0 || L >= 32 || S < 0 || S > 8 || !one_of(pl.path, {0, 1, 2, 3, 4}) || !one_of(pl.P, {1, 2, 4, 8}) ||
        !one_of(pl.pad, {0, 160 * 1024 / 6, 160 * 1024 / 4}) || !one_of(pl.splitk, {0, 4, 8}) || !one_of(pl.pp, {1, 2, 4}) ||
        !one_of(pl.w16, {0, 1}) || !one_of(pl.fuse, {0, 1}) || !one_of(pl.hiacc, {0, 1}) || !one_of(pl.ks, {0, 2, 4, 8, 16}))
        return false;
    *key = {B, {L, S}};
    *out = pl;
    return true;
}

namespace {
struct PlanTable {
    std::map<Y2PlanKey, Y2PlanLine> lines;
    std::map<int, int> per_batch;
};
std::mutex g_tables_mu;
std::map<std::string, std::shared_ptr<PlanTable>> g_tables;   // by file name: loaded once each

// config/plan_gfx950.txt of the PACKAGE the library belongs to: next to the library, or - for an A/B variant that
// tools/build_variant.sh links into <package>/build/ - one directory up (variants silently ran the autotune while the
// baseline ran the table, so A/B numbers mixed the plan source with the variant under test).
std::string default_plan_path()
{
    Dl_info info;
    if (!dladdr((const void *)&y2_plan_line_parse, &info) || !info.dli_fname) return "";
    std::string lib = info.dli_fname;
    const size_t slash = lib.rfind('/');
    const std::string dir = slash == std::string::npos ? std::string(".") : lib.substr(0, slash);
    for (const std::string &cand : {dir + "/config/plan_gfx950.txt", dir + "/../config/plan_gfx950.txt"}) {
        FILE *f = fopen(cand.c_str(), "r");
        if (f) { fclose(f); return cand; }
    }
    return "";
}

std::shared_ptr<PlanTable> table_for(const Y2Options &opt)
{
    const std::string path = !opt.plan_file.empty() ? opt.plan_file : default_plan_path();
    std::lock_guard<std::mutex> lk(g_tables_mu);
    auto it = g_tables.find(path);
    if (it != g_tables.end()) return it->second;
    auto t = std::make_shared<PlanTable>();
    FILE *f = path.empty() ? nullptr : fopen(path.c_str(), "r");
    if (f) {
        char line[256];
        while (fgets(line, sizeof(line), f)) {
            Y2PlanKey k;
            Y2PlanLine pl;
            // a line outside what the planner itself can produce is dropped here (its batch then misses a launch and is timed)
            if (!y2_plan_line_parse(line, &k, &pl)) continue;
            if (!t->lines.count(k)) t->per_batch[k.first]++;
            t->lines[k] = pl;       // a later line for the same key wins (appended re-measurements)
        }
        fclose(f);
    }
    g_tables[path] = t;
    return t;
}
}  // namespace

bool y2_plan_table_has_batch(const Y2Options &opt, int B) { return table_for(opt)->per_batch.count(B) != 0; }

bool y2_plan_table_lookup(const Y2Options &opt, int B, int L, int S, Y2PlanLine *out)
{
    auto t = table_for(opt);
    auto it = t->lines.find({B, {L, S}});
    if (it == t->lines.end()) return false;
    *out = it->second;
    return true;
}

// ---------------------------------------------------------------------------- the weight-side cache

uint64_t y2_hash_bytes(uint64_t h, const void *data, size_t n)   // FNV-1a 64 (small inputs: Q tables, the cache file's body)
{
    const unsigned char *p = (const unsigned char *)data;
    if (!h) h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

static const char kCacheMagic[] = "Y2PLAN 2 gfx950";   // format 2; another architecture's plans are another file's business

bool Y2PlanCache::load(uint64_t want_hash, std::string *why)
{
    std::lock_guard<std::mutex> lk(mu);
    lines.clear();
    per_batch.clear();
    bounds_valid = false;
    dirty = false;
    hash = want_hash;
    for (auto &b : bounds) b = Bounds();
    auto no = [&](const char *w) { if (why) *why = w; lines.clear(); per_batch.clear(); for (auto &b : bounds) b = Bounds(); return false; };
    FILE *f = path.empty() ? nullptr : fopen(path.c_str(), "r");
    if (!f) return no("no cache file");
    std::string body;
    char line[4096];
    bool magic = false, hash_ok = false, sum_seen = false;
    int n_bounds = 0;
    while (fgets(line, sizeof(line), f)) {
        if (sum_seen) { fclose(f); return no("data after the checksum line"); }
        if (!strncmp(line, "sum ", 4)) {
            unsigned long long s = 0;
            if (sscanf(line + 4, "%llx", &s) != 1 || s != y2_hash_bytes(0, body.data(), body.size())) { fclose(f); return no("checksum mismatch (damaged file)"); }
            sum_seen = true;
            continue;
        }
        body += line;
        if (!magic) {
            if (strncmp(line, kCacheMagic, sizeof(kCacheMagic) - 1)) { fclose(f); return no("not a plan cache of this format / architecture"); }
            magic = true;
            continue;
        }
        if (!strncmp(line, "hash ", 5)) {
            unsigned long long h = 0;
            if (sscanf(line + 5, "%llx", &h) != 1 || h != want_hash) { fclose(f); return no("made for another weight set (hash differs)"); }
            hash_ok = true;
            continue;
        }
        if (!strncmp(line, "bound ", 6)) {
            int ord, ms, mb, MB, used = 0;
            if (sscanf(line + 6, "%d %d %d %d%n", &ord, &ms, &mb, &MB, &used) != 4 || ord < 0 || ord >= YOLO2_N_CONV || MB <= 0 || MB > 64) { fclose(f); return no("bad bound line"); }
            Bounds &b = bounds[ord];
            b.maxsum = ms; b.maxbias = mb;
            const char *p = line + 6 + used;
            for (int k = 0; k < MB; ++k) {
                int v[5], u = 0;
                if (sscanf(p, "%d %d %d %d %d%n", &v[0], &v[1], &v[2], &v[3], &v[4], &u) != 5) { fclose(f); return no("bad bound line"); }
                b.sum_mb.push_back(v[0]); b.bias_mb.push_back(v[1]); b.abs_mb.push_back(v[2]); b.form.push_back(v[3]); b.scale.push_back(v[4]);
                p += u;
            }
            ++n_bounds;
            continue;
        }
        if (!strncmp(line, "plan ", 5)) {
            Y2PlanKey k;
            Y2PlanLine pl;
            if (!y2_plan_line_parse(line + 5, &k, &pl)) { fclose(f); return no("bad plan line"); }
            if (!lines.count(k)) per_batch[k.first]++;
            lines[k] = pl;
            continue;
        }
        if (line[0] == '#' || line[0] == '\n') continue;
        fclose(f);
        return no("unknown line");
    }
    fclose(f);
    if (!magic || !hash_ok || !sum_seen) return no("truncated file");
    if (n_bounds != YOLO2_N_CONV) return no("bounds of some conv layers are missing");
    bounds_valid = true;
    return true;
}

bool Y2PlanCache::save()
{
    std::lock_guard<std::mutex> lk(mu);
    if (path.empty() || !hash) return false;
    std::string body = std::string(kCacheMagic) + "\n";
    char buf[256];
    body += "# weight-side plan cache of libyolo2_hip.so: bounds behind the arithmetic-form proofs, forms / scale shifts per block of 32\n"
            "# output channels, and the timed conv plan per batch (same fields as config/plan_gfx950.txt).  Delete it to re-time.\n";
    snprintf(buf, sizeof(buf), "hash %016llx\n", (unsigned long long)hash);
    body += buf;
    for (int o = 0; o < YOLO2_N_CONV; ++o) {
        const Bounds &b = bounds[o];
        snprintf(buf, sizeof(buf), "bound %d %d %d %zu", o, b.maxsum, b.maxbias, b.sum_mb.size());
        body += buf;
        for (size_t k = 0; k < b.sum_mb.size(); ++k) {
            snprintf(buf, sizeof(buf), " %d %d %d %d %d", b.sum_mb[k], b.bias_mb[k], b.abs_mb[k], b.form[k], b.scale[k]);
            body += buf;
        }
        body += "\n";
    }
    for (const auto &kv : lines) {
        const Y2PlanLine &p = kv.second;
        snprintf(buf, sizeof(buf), "plan %d %d %d %d %d %d %d %d %d %d %d %d\n", kv.first.first, kv.first.second.first, kv.first.second.second, p.path, p.P,
                 p.pad, p.splitk, p.pp, p.w16, p.fuse, p.hiacc, p.ks);
        body += buf;
    }
    snprintf(buf, sizeof(buf), "sum %016llx\n", (unsigned long long)y2_hash_bytes(0, body.data(), body.size()));
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "w");
    if (!f) return false;      // read-only weight directory: the cache is an optimisation, not a requirement
    const bool ok = fwrite(body.data(), 1, body.size(), f) == body.size() && fputs(buf, f) >= 0;
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) { remove(tmp.c_str()); return false; }
    dirty = false;
    return true;
}

// Binds a cache file to the context (NULL or "" unbinds).  Takes effect at the next weight load.
extern "C" int yolo2_hip_set_plan_cache(yolo2_hip_ctx *c, const char *path)
{
    if (!c) return fail(YOLO2_ERROR, "null ctx");
    if (c->is_lane) return fail(YOLO2_ERROR, "lanes share their parent's plan cache");
    if (!path || !path[0]) { c->plan_cache.reset(); return YOLO2_SUCCESS; }
    c->plan_cache = std::make_shared<Y2PlanCache>();
    c->plan_cache->path = path;
    return YOLO2_SUCCESS;
}

// What the context knows about its cache: hash of the loaded weight set, whether the file's bounds were used at the last load,
// how many plan lines / batches it holds.  YOLO2_ERROR if no cache is bound.
extern "C" int yolo2_hip_plan_cache_info(yolo2_hip_ctx *c, uint64_t *weights_hash, int *bounds_from_file, int *n_lines, int *n_batches)
{
    if (!c || !c->plan_cache) return fail(YOLO2_ERROR, "no plan cache bound (yolo2_hip_set_plan_cache)");
    std::lock_guard<std::mutex> lk(c->plan_cache->mu);
    if (weights_hash) *weights_hash = c->plan_cache->hash;
    if (bounds_from_file) *bounds_from_file = c->plan_cache->bounds_valid ? 1 : 0;
    if (n_lines) *n_lines = (int)c->plan_cache->lines.size();
    if (n_batches) *n_batches = (int)c->plan_cache->per_batch.size();
    return YOLO2_SUCCESS;
}

// Test hook (no GPU): would the file at `path` be accepted for a weight set with this hash?  YOLO2_SUCCESS, or YOLO2_ERROR with the
// reason in yolo2_hip_last_error() - a stale or damaged file is refused as a whole, the loader then recomputes and re-times.
extern "C" int yolo2_hip_plan_cache_check(const char *path, uint64_t weights_hash, int *n_lines)
{
    if (!path) return fail(YOLO2_ERROR, "null path");
    Y2PlanCache pc;
    pc.path = path;
    std::string why;
    if (!pc.load(weights_hash, &why)) return fail(YOLO2_ERROR, "plan cache %s refused: %s", path, why.c_str());
    if (n_lines) *n_lines = (int)pc.lines.size();
    return YOLO2_SUCCESS;
}
