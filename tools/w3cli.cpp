// w3cli — C++ host above the C ABI (include/w3hip.h), mirroring the reference
// binary's CLI (src/main.rs:24-87): `w3 <c|d|t> <path>`; output goes to the
// current directory as <name>.bin (compress/test) or <name>.orig (decompress);
// a directory is traversed shallowly (main.rs:41-50).
//
// Container: by default the block container of DESIGN.md §3 ("w3bk"), which is
// what the GPU path is for.  `W3_CONTAINER=w30i` selects the reference's own
// single-stream container (main.rs:14-15,95-96) — one GPU lane, format parity.
// Model: init_model() of main.rs:151 — OrderNEntropy(11,3,ACHistory(8,book1)).
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <deque>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <vector>

#include "../include/w3hip.h"

static const uint32_t kBlock = 65536;

static void push(w3_model_spec &s, uint8_t kind, uint8_t bits = 0, uint8_t align = 0, uint8_t max_bits = 0, uint8_t log_cells = 0) {
    w3_node &n = s.nodes[s.n_nodes++];
    n.kind = kind; n.bits = bits; n.align = align; n.max_bits = max_bits; n.log_cells = log_cells;
}

// The reference chooses its model at compile time in init_model() (main.rs:146-152); here W3_MODEL picks one:
//   (unset) / default : main.rs:151, OrderNEntropy(11, 3, ACHistory(8, StationaryModel::for_book1()))
//   order012          : the commented alternative of main.rs:148-150 in today's types, BestOfTwo(BestOfTwo(Order0, Order1), OrderN(27,3))
//   order012apm       : + one APM stage (BASELINE configs[1]; build-defined)
//   fullcm            : + slot-state leaves of order 1-4 and two APM stages (BASELINE configs[2])
static w3_model_spec init_model() {
    w3_model_spec s;
    memset(&s, 0, sizeof s);
    const char *m = getenv("W3_MODEL");
    const std::string name = m ? m : "default";
    if (name == "default") {
        push(s, W3_NODE_ORDERN, 11, 3, 8);
        s.nodes[0].history = W3_HIST_AC;
        const uint16_t book1[8] = {1, 50188, 62497, 15819, 22545, 31499, 22988, 29616};  // stationary.rs:41
        memcpy(s.nodes[0].table, book1, sizeof book1);
        return s;
    }
    if (name != "order012" && name != "order012apm" && name != "fullcm") { fprintf(stderr, "unknown W3_MODEL %s\n", name.c_str()); exit(1); }
    push(s, W3_NODE_ORDERN, 11, 3); push(s, W3_NODE_ORDERN, 19, 3); push(s, W3_NODE_BEST_OF_TWO);
    push(s, W3_NODE_ORDERN, 27, 3); push(s, W3_NODE_BEST_OF_TWO);
    if (name == "fullcm")
        for (uint8_t order = 1; order <= 4; order++) { push(s, W3_NODE_SLOT_STATE, order, 0, 0, 14); push(s, W3_NODE_BEST_OF_TWO); }
    if (name != "order012") push(s, W3_NODE_APM, 0, W3_APM_ORDER0, 7);
    if (name == "fullcm") push(s, W3_NODE_APM, 0, W3_APM_ORDER1, 6);
    return s;
}

static bool read_file(const std::string &p, std::vector<uint8_t> &out) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}
static bool write_file(const std::string &p, const uint8_t *d, size_t n) {
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) return false;
    bool ok = n == 0 || fwrite(d, 1, n, f) == n;
    fclose(f);
    return ok;
}
static void put_be(std::vector<uint8_t> &v, uint64_t x, int bytes) { for (int i = bytes - 1; i >= 0; i--) v.push_back((uint8_t)(x >> (8 * i))); }
static uint64_t get_be(const uint8_t *p, int bytes) { uint64_t x = 0; for (int i = 0; i < bytes; i++) x = (x << 8) | p[i]; return x; }

static std::string out_path(const std::string &in, const char *ext) {  // main.rs:58-68
    size_t slash = in.find_last_of('/');
    std::string name = slash == std::string::npos ? in : in.substr(slash + 1);
    size_t dot = name.find_last_of('.');
    if (dot != std::string::npos && dot != 0) name = name.substr(0, dot);
    return name + "." + ext;
}

static int die(w3_ctx *ctx, int rc, const char *what) {
    fprintf(stderr, "%s: %s (%s)\n", what, w3_strerror(rc), ctx ? w3_last_error(ctx) : "");
    return 1;
}

static bool write_block_container(const std::string &out, size_t orig, const std::vector<uint32_t> &lens, size_t nb, const uint8_t *body, size_t blen);
// bytes per w3_encode_blocks / w3_decode_blocks call: a multiple of the block size below the library's 4 GiB limit
static size_t call_max() {
    size_t m = (size_t)1 << 31;
    if (const char *e = getenv("W3_CALL_MAX")) { const long long v = atoll(e); if (v > 0) m = (size_t)v; }
    return std::max<size_t>(kBlock, m / kBlock * kBlock);
}

static int compress(w3_ctx *ctx, const std::string &in, const std::string &out) {
    std::vector<uint8_t> data;
    if (!read_file(in, data)) { perror(in.c_str()); return 1; }
    w3_model_spec spec = init_model();
    const char *cont = getenv("W3_CONTAINER");
    if (cont && !strcmp(cont, "w30i")) {
        std::vector<uint8_t> buf(2 * data.size() + 128);
        size_t len = 0;
        int rc = w3_compress_stream(ctx, &spec, data.data(), data.size(), buf.data(), buf.size(), &len);
        if (rc) return die(ctx, rc, "w3_compress_stream");
        return write_file(out, buf.data(), len) ? 0 : 1;
    }
    size_t nb = (data.size() + kBlock - 1) / kBlock;
    std::vector<uint8_t> body(2 * data.size() + 64 * nb + 64);
    std::vector<uint32_t> lens(nb ? nb : 1);
    size_t blen = 0;
    int rc;
    // W3_SHARDS=k: the blocks as k contiguous ranges on k contexts, one per GPU (round robin over the devices present) —
    // w3_encode_blocks_sharded, the single-process form of BASELINE configs[3]; same container bytes as one context
    const char *sh = getenv("W3_SHARDS");
    const int k = sh ? atoi(sh) : 1;
    if (k > 1 && k <= 64) {
        int ndev = 1;
        const char *nd = getenv("W3_NDEV");
        if (nd && atoi(nd) > 0) ndev = atoi(nd);
        std::vector<w3_ctx *> cs((size_t)k, nullptr);
        cs[0] = ctx;
        for (int r = 1; r < k; r++)
            if (w3_ctx_create(r % ndev, &cs[r]) != W3_OK) { fprintf(stderr, "w3_ctx_create(device %d) failed\n", r % ndev); return 1; }
        rc = w3_encode_blocks_sharded(cs.data(), k, &spec, data.data(), data.size(), kBlock, body.data(), body.size(), &blen, lens.data());
        if (rc == W3_E_NOSPACE) { body.resize(blen); rc = w3_encode_blocks_sharded(cs.data(), k, &spec, data.data(), data.size(), kBlock, body.data(), body.size(), &blen, lens.data()); }
        for (int r = 1; r < k; r++) w3_ctx_destroy(cs[r]);
        if (rc) return die(ctx, rc, "w3_encode_blocks_sharded");
    } else {
        // one call handles less than 4 GiB (w3hip.h): a larger file goes through in pieces of whole blocks — blocks are independent, so the
        // container is the same bytes whatever the pieces (W3_CALL_MAX=<bytes>: the piece size, for tests)
        const size_t piece = call_max();
        for (size_t o = 0; o < data.size() || o == 0; o += piece) {
            const size_t n_p = std::min(piece, data.size() - o), b0 = o / kBlock;
            size_t len_p = 0;
            rc = w3_encode_blocks(ctx, &spec, data.data() + o, n_p, kBlock, body.data() + blen, body.size() - blen, &len_p, lens.data() + b0);
            if (rc == W3_E_NOSPACE) { body.resize(blen + len_p + (data.size() - o)); rc = w3_encode_blocks(ctx, &spec, data.data() + o, n_p, kBlock, body.data() + blen, body.size() - blen, &len_p, lens.data() + b0); }
            if (rc) return die(ctx, rc, "w3_encode_blocks");
            blen += len_p;
            if (data.empty()) break;
        }
    }
    return write_block_container(out, data.size(), lens, nb, body.data(), blen) ? 0 : 1;
}

static bool write_block_container(const std::string &out, size_t orig, const std::vector<uint32_t> &lens, size_t nb, const uint8_t *body, size_t blen) {
    std::vector<uint8_t> file = {'w', '3', 'b', 'k', 1};
    put_be(file, orig, 8); put_be(file, kBlock, 4); put_be(file, nb, 4);
    for (size_t b = 0; b < nb; b++) put_be(file, lens[b], 4);
    file.insert(file.end(), body, body + blen);
    return write_file(out, file.data(), file.size());
}

// A DIRECTORY compressed with FILES IN FLIGHT (main.rs:41-50 walks it one file after the other): w3_encode_host_submit takes file k+1 —
// its input crosses PCIe and its encode is enqueued — while file k is still being coded and file k-1's streams travel back; a file's
// coder chain (8 x 65,536 dependent steps per lane, ~17 ms however small the file) overlaps the other files' instead of being waited
// for.  Same container bytes as the one-file-at-a-time path.
static int compress_dir_in_flight(w3_ctx *ctx, const std::vector<std::string> &files) {
    struct Pending { std::string in, out; std::vector<uint8_t> data, body; std::vector<uint32_t> lens; size_t nb = 0; int hjob = -1;
                     std::chrono::steady_clock::time_point t0; };
    std::deque<std::unique_ptr<Pending>> q;
    w3_model_spec spec = init_model();
    int ret = 0;
    auto finish = [&]() {
        std::unique_ptr<Pending> p = std::move(q.front());
        q.pop_front();
        size_t blen = 0;
        int rc = w3_encode_host_wait(ctx, p->hjob, &blen);
        if (rc == W3_E_NOSPACE) {   // (a file that expands past 2 n: once more, alone, with the room it asked for)
            p->body.resize(blen);
            rc = w3_encode_blocks(ctx, &spec, p->data.data(), p->data.size(), kBlock, p->body.data(), p->body.size(), &blen, p->lens.data());
        }
        if (rc) { ret |= die(ctx, rc, "w3_encode_host_wait"); return; }
        if (!write_block_container(p->out, p->data.size(), p->lens, p->nb, p->body.data(), blen)) { ret |= 1; return; }
        printf("Compression took: %.3fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - p->t0).count());
    };
    for (const std::string &f : files) {
        std::unique_ptr<Pending> p(new Pending);
        p->in = f; p->out = out_path(f, "bin"); p->t0 = std::chrono::steady_clock::now();
        if (!read_file(f, p->data)) { perror(f.c_str()); ret |= 1; continue; }
        if (p->data.empty()) {   // (nothing to submit: an empty block container)
            ret |= write_block_container(p->out, 0, p->lens, 0, nullptr, 0) ? 0 : 1;
            printf("Compression took: %.3fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - p->t0).count());
            continue;
        }
        p->nb = (p->data.size() + kBlock - 1) / kBlock;
        p->body.resize(2 * p->data.size() + 64 * p->nb + 64);
        p->lens.resize(p->nb);
        while (!q.empty() && (int)q.size() >= w3_encode_host_max_in_flight(&spec, p->data.size(), kBlock)) finish();
        int rc = w3_encode_host_submit(ctx, &spec, p->data.data(), p->data.size(), kBlock, p->body.data(), p->body.size(), p->lens.data(), &p->hjob);
        if (rc == W3_E_INVALID && !q.empty()) {   // (a larger file allows fewer calls in flight than the ones before it: drain, then retry)
            while (!q.empty()) finish();
            rc = w3_encode_host_submit(ctx, &spec, p->data.data(), p->data.size(), kBlock, p->body.data(), p->body.size(), p->lens.data(), &p->hjob);
        }
        if (rc) { ret |= die(ctx, rc, "w3_encode_host_submit"); continue; }
        q.push_back(std::move(p));
    }
    while (!q.empty()) finish();
    return ret;
}

static int decompress(w3_ctx *ctx, const std::string &in, const std::string &out) {
    std::vector<uint8_t> data;
    if (!read_file(in, data)) { perror(in.c_str()); return 1; }
    w3_model_spec spec = init_model();
    if (data.size() >= 4 && !memcmp(data.data(), "w30i", 4)) {
        if (data.size() < 12) return die(ctx, W3_E_FORMAT, "header");
        std::vector<uint8_t> o((size_t)get_be(data.data() + 4, 8) + 1);
        size_t len = 0;
        int rc = w3_decompress_stream(ctx, &spec, data.data(), data.size(), o.data(), o.size(), &len);
        if (rc) return die(ctx, rc, "w3_decompress_stream");
        return write_file(out, o.data(), len) ? 0 : 1;
    }
    if (data.size() < 21 || memcmp(data.data(), "w3bk", 4) || data[4] != 1) {  // main.rs:123-124 asserts the magic
        fprintf(stderr, "Magic numbers don't match up - file wasn't compressed with (this version of) w3cli!\n");
        return 1;
    }
    uint64_t orig = get_be(data.data() + 5, 8);
    uint32_t bs = (uint32_t)get_be(data.data() + 13, 4), nb = (uint32_t)get_be(data.data() + 17, 4);
    if (data.size() < 21 + 4ull * nb) return die(ctx, W3_E_FORMAT, "length table");
    std::vector<uint32_t> lens(nb ? nb : 1);
    uint64_t total = 0;
    for (uint32_t b = 0; b < nb; b++) { lens[b] = (uint32_t)get_be(data.data() + 21 + 4ull * b, 4); total += lens[b]; }
    if (data.size() < 21 + 4ull * nb + total) return die(ctx, W3_E_FORMAT, "streams");
    std::vector<uint8_t> o((size_t)orig + 1);
    const uint8_t *body = data.data() + 21 + 4ull * nb;
    const size_t per_call = bs ? std::max<size_t>(1, call_max() / bs) : nb;   // blocks per call (compress() above)
    uint64_t coff = 0;
    for (size_t b0 = 0; b0 < nb || b0 == 0; b0 += per_call) {
        const size_t b1 = std::min<size_t>(nb, b0 + per_call);
        uint64_t clen = 0;
        for (size_t b = b0; b < b1; b++) clen += lens[b];
        const uint64_t o0 = (uint64_t)b0 * bs, o1 = std::min<uint64_t>(orig, (uint64_t)b1 * bs);
        int rc = w3_decode_blocks(ctx, &spec, body + coff, (size_t)clen, lens.data() + b0, b1 - b0, bs, (size_t)(o1 > o0 ? o1 - o0 : 0), o.data() + o0);
        if (rc) return die(ctx, rc, "w3_decode_blocks");
        coff += clen;
        if (nb == 0) break;
    }
    return write_file(out, o.data(), (size_t)orig) ? 0 : 1;
}

static int run(w3_ctx *ctx, const std::string &path, char action) {  // main.rs:55-87
    auto t0 = std::chrono::steady_clock::now();
    int rc;
    if (action == 'c') { rc = compress(ctx, path, out_path(path, "bin")); if (!rc) printf("Compression took: %.3fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
    else if (action == 'd') { rc = decompress(ctx, path, out_path(path, "orig")); if (!rc) printf("Decompression took: %.3fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
    else { rc = run(ctx, path, 'c'); if (!rc) rc = run(ctx, out_path(path, "bin"), 'd'); }
    return rc;
}

static void usage(const char *msg) {  // main.rs:154-161
    printf("Usage: w3 <Action> <Path>\n<Action> [single file]: c (compress), d (decompress), t (test = c + d)\n"
           "<Path> can be a single file or a directory\nNote: Directories are shallow traversed\n\n%s\n", msg);
    exit(1);
}

int main(int argc, char **argv) {
    if (argc != 3) usage("Invokation doesn't match usage! Provide 2 arguments.");
    char action = argv[1][0];
    if (strlen(argv[1]) != 1 || (action != 'c' && action != 'd' && action != 't')) usage("Unrecognized option -> <action>!");
    struct stat st;
    if (stat(argv[2], &st)) { fprintf(stderr, "Path must be a file or a directory!\n"); return 1; }
    w3_ctx *ctx = nullptr;
    int rc = w3_ctx_create(0, &ctx);
    if (rc) return die(nullptr, rc, "w3_ctx_create (an MI355X is required; there is no CPU path)");
    int ret = 0;
    if (S_ISDIR(st.st_mode)) {
        std::vector<std::string> files;
        DIR *d = opendir(argv[2]);
        while (dirent *e = d ? readdir(d) : nullptr) {
            std::string p = std::string(argv[2]) + "/" + e->d_name;
            struct stat s2;
            if (!stat(p.c_str(), &s2) && S_ISREG(s2.st_mode)) files.push_back(p);
        }
        if (d) closedir(d);
        const char *cont = getenv("W3_CONTAINER"), *sh = getenv("W3_SHARDS"), *serial = getenv("W3_SERIAL");
        const bool block_container = !(cont && !strcmp(cont, "w30i")) && !(sh && atoi(sh) > 1);
        if (action == 'c' && block_container && !serial) ret = compress_dir_in_flight(ctx, files);   // files in flight (W3_SERIAL=1: one after the other)
        else for (const std::string &p : files) ret |= run(ctx, p, action);
    } else {
        ret = run(ctx, argv[2], action);
    }
    w3_ctx_destroy(ctx);
    return ret;
}
