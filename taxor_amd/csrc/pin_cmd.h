// pin_cmd.h -- `taxor pin`: ONE command that turns a published .hixf + a query file + the TSV the reference wrote for them into
// a committed parity fixture (tests/golden/real_<name>.json), SURVEY.md 8(f) #2.  Included by search_main.cpp inside its
// anonymous namespace (it uses that file's helpers: die, file_exists, str_split, parse_arith_spec, arith_spec, fnv1a,
// compare_tsv, variant_family, rank_variants).
//
//   taxor pin --index-file X.hixf --query-file R.fq --expect ref.tsv --out tests/golden/real_<name>.json
//             [--ixf-arithmetic spec] [--ixf-layout spec] [--error-rate e] [--percentage p] [--threads n] [--fixture-reads n] [--gpu id] [--name s]
//
// What it does, in order:
//   1. probe     : the IXF record layout of the file (taxor_hixf_probe), loaded with what was found;
//   2. variants  : which reading of the un-vendored IXF the file follows -- arithmetic AND fingerprint layout: the root IXF's raw
//                  bytes are probed under the variant family of `taxor verify --variants`, with hash lists cut from the query
//                  reads the EXPECTED TSV reports as the best matches (no genome file needed: the reference's own output says
//                  which reads are in the index); skipped when --ixf-arithmetic names the arithmetic (the layout is then what
//                  --ixf-layout says, or what the loader settled on);
//   3. search    : `taxor search --ixf-arithmetic <that> --ixf-layout <that> --expect ref.tsv` as a child process (a layout
//                  other than the search layout is transposed on the device at load), every read compared with the
//                  reference's lines (matched by id, a read's lines in DFS order);
//   4. fixture   : schema, arithmetic code, the comparison's counts and, for a handful of reads, everything a test needs to
//                  re-check the un-vendored boundaries WITHOUT the (hundreds of megabytes of) index: the read itself, the
//                  reference's lines for it, its distinct syncmer hashes, and for every IXF on the path to its reported bins
//                  the three probe rows of every hash plus the three fingerprint bytes found there for each technical bin of
//                  the reported user bin (and of the merged bins above it).  tests/real_fixture_check.py recomputes hashes,
//                  rows, fingerprints and match counts with the CPU oracle and with the HIP path and holds them against the
//                  REFERENCE's QHASH_COUNT / QHASH_MATCH -- the oracle pinned by the reference's own output.
// Exit code 0 = every read identical (fixture says "pinned": true), 3 = differences (the fixture is still written, for diagnosis).

struct PinRead {
    std::string id, seq;
    std::vector<std::string> expect;      // the reference's lines for this read
};

static std::string json_escape(const std::string &s)
{
    std::string o;
    o.reserve(s.size() + 8);
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back((char)c); }
        else if (c == '\n') o += "\\n";
        else if (c == '\t') o += "\\t";
        else if (c == '\r') o += "\\r";
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o.push_back((char)c);
    }
    return o;
}

static std::string b64(const void *p, size_t n)
{
    static const char *T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    const unsigned char *d = (const unsigned char *)p;
    std::string o;
    o.reserve((n + 2) / 3 * 4);
    for (size_t i = 0; i < n; i += 3) {
        const uint32_t v = (uint32_t)d[i] << 16 | (i + 1 < n ? (uint32_t)d[i + 1] << 8 : 0u) | (i + 2 < n ? (uint32_t)d[i + 2] : 0u);
        o.push_back(T[v >> 18 & 63]);
        o.push_back(T[v >> 12 & 63]);
        o.push_back(i + 1 < n ? T[v >> 6 & 63] : '=');
        o.push_back(i + 2 < n ? T[v & 63] : '=');
    }
    return o;
}

static std::vector<std::string> split_tabs(const std::string &l)
{
    std::vector<std::string> f;
    size_t a = 0;
    for (;;) {
        const size_t b = l.find('\t', a);
        if (b == std::string::npos) { f.push_back(l.substr(a)); break; }
        f.push_back(l.substr(a, b - a));
        a = b + 1;
    }
    return f;
}

static int pin_command(int argc, char **argv)
{
    std::string index_file, query_file, expect_file, out_file, name, arith_given, layout_given;
    double error_rate = 0.04, percentage = -1.0;
    unsigned threads = 8;
    size_t fixture_reads = 8;
    int device = 0;
    for (int i = 2; i < argc; ++i) {
        auto val = [&]() -> std::string { if (i + 1 >= argc) die(std::string("Missing value for option ") + argv[i]); return argv[++i]; };
        const std::string k = argv[i];
        if (k == "--index-file" || k == "--index") index_file = val();
        else if (k == "--query-file" || k == "--query") query_file = val();
        else if (k == "--expect") expect_file = val();
        else if (k == "--out") out_file = val();
        else if (k == "--name") name = val();
        else if (k == "--ixf-arithmetic") arith_given = val();
        else if (k == "--ixf-layout") layout_given = val();
        else if (k == "--error-rate") error_rate = atof(val().c_str());
        else if (k == "--percentage") percentage = atof(val().c_str());
        else if (k == "--threads") threads = (unsigned)atoi(val().c_str());
        else if (k == "--fixture-reads") fixture_reads = (size_t)strtoull(val().c_str(), nullptr, 10);
        else if (k == "--gpu") device = atoi(val().c_str());
        else die("taxor pin: unknown option " + k);
    }
    if (index_file.empty() || query_file.empty() || expect_file.empty() || out_file.empty())
        die("usage: taxor pin --index-file <x.hixf> --query-file <reads.fq> --expect <reference.tsv> --out tests/golden/real_<name>.json\n"
            "                 [--ixf-arithmetic kh=..,sm=..,rot=..,red=..,fp=..] [--ixf-layout bin-major,unpadded,..] [--error-rate e] [--percentage p] [--threads n] [--fixture-reads n]");
    for (const auto &f : {index_file, query_file, expect_file})
        if (!file_exists(f)) die("taxor pin: no such file: " + f);
    if (threads < 1 || threads > 32) threads = 8;
    if (name.empty()) {
        name = out_file.substr(out_file.find_last_of('/') == std::string::npos ? 0 : out_file.find_last_of('/') + 1);
        if (name.rfind("real_", 0) == 0) name = name.substr(5);
        if (name.size() > 5 && name.substr(name.size() - 5) == ".json") name.resize(name.size() - 5);
    }

    // ---- 1. probe + load ----------------------------------------------------------------------------------------------------
    taxor_ixf_schema schema;
    std::vector<char> rep(16384);
    if (taxor_hixf_probe(index_file.c_str(), &schema, rep.data(), rep.size()) != TAXOR_OK) die(taxor_gpu_last_error());
    printf("== probe\n%s", rep.data());
    taxor_hixf *h = nullptr;
    if (taxor_hixf_load_schema(index_file.c_str(), &schema, &h) != TAXOR_OK) die(taxor_gpu_last_error());
    const taxor_hixf_view *view = taxor_hixf_get_view(h);
    const taxor_hixf_meta *meta = taxor_hixf_get_meta(h);
    if (!view->use_syncmer) die("taxor pin: the index was built without --use-syncmer; the fixture format covers syncmer indexes (the BASELINE configs)");

    // ---- the reference's TSV: lines per read, and the reads it reports as the best matches -----------------------------------
    std::map<std::string, std::vector<std::string>> expect;
    std::vector<std::string> expect_order;
    {
        FILE *f = fopen(expect_file.c_str(), "rb");
        if (!f) die("cannot open " + expect_file);
        std::string text;
        std::vector<char> buf(1 << 20);
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) text.append(buf.data(), n);
        fclose(f);
        size_t a = 0;
        while (a < text.size()) {
            size_t b = text.find('\n', a);
            if (b == std::string::npos) b = text.size();
            std::string l = text.substr(a, b - a);
            if (!l.empty() && l.back() == '\r') l.pop_back();
            a = b + 1;
            if (l.empty() || l[0] == '#') continue;
            const std::string id = l.substr(0, l.find('\t'));
            auto it = expect.find(id);
            if (it == expect.end()) { it = expect.emplace(id, std::vector<std::string>()).first; expect_order.push_back(id); }
            it->second.push_back(l);
        }
    }
    if (expect.empty()) die("taxor pin: " + expect_file + " holds no result lines");
    struct Scored { double ratio; std::string id; };
    std::vector<Scored> scored;
    std::vector<std::string> classified, unclassified;
    for (const auto &id : expect_order) {
        const auto f = split_tabs(expect[id][0]);
        if (f.size() >= 10 && f[1] != "-") {
            classified.push_back(id);
            double best = 0;
            for (const auto &l : expect[id]) {
                const auto g = split_tabs(l);
                const double c = g.size() >= 8 ? atof(g[6].c_str()) : 0.0, m = g.size() >= 8 ? atof(g[7].c_str()) : 0.0;
                if (c >= 50) best = std::max(best, m / c);
            }
            if (best > 0) scored.push_back({best, id});
        } else
            unclassified.push_back(id);
    }
    std::sort(scored.begin(), scored.end(), [](const Scored &a, const Scored &b) { return a.ratio != b.ratio ? a.ratio > b.ratio : a.id < b.id; });
    if (scored.size() > 24) scored.resize(24);
    std::map<std::string, PinRead> wanted;         // reads whose sequences are needed: positives of the scan + the fixture's
    for (const auto &s : scored) wanted[s.id];
    std::vector<std::string> fixture_ids;
    {
        const size_t n_un = std::min<size_t>(unclassified.size(), fixture_reads >= 4 ? 2 : 0);
        const size_t n_cl = std::min(classified.size(), fixture_reads - n_un);
        for (size_t i = 0; i < n_cl; ++i) fixture_ids.push_back(classified[i * classified.size() / n_cl]);      // spread over the file
        for (size_t i = 0; i < n_un; ++i) fixture_ids.push_back(unclassified[i * unclassified.size() / n_un]);
        for (const auto &id : fixture_ids) wanted[id];
    }
    {
        fastx::FastxReader rd;
        std::string id, seq;
        size_t found = 0;
        try {
            for (const auto &qf : str_split(query_file, ',')) {
                if (!rd.open(qf)) die("cannot open " + qf);
                for (;;) {
                    seq.clear();
                    if (!rd.next(id, seq)) break;
                    auto it = wanted.find(id);
                    if (it == wanted.end()) {                         // the reference may have written the id up to the first blank
                        const size_t sp = id.find_first_of(" \t");
                        if (sp != std::string::npos) it = wanted.find(id.substr(0, sp));
                    }
                    if (it != wanted.end() && it->second.seq.empty()) {
                        it->second.id = it->first;
                        it->second.seq = seq;
                        it->second.expect = expect[it->first];
                        ++found;
                    }
                }
            }
        } catch (const std::exception &e) { die(e.what()); }
        if (found < wanted.size())
            fprintf(stderr, "[taxor pin] %zu of the %zu reads picked from the expected TSV were not found in the query file by id\n", wanted.size() - found, wanted.size());
    }

    // ---- 2. the reading the file follows: arithmetic and fingerprint layout --------------------------------------------------
    // hashing needs k / s / t only: a one-IXF dummy index carries them; the scan itself probes the ROOT's raw bytes (every indexed
    // genome is in one of its bins), which keeps this step cheap for a 113 GB index
    if (!layout_given.empty()) {
        uint32_t lc = 0;
        if (taxor_ixf_layout_parse(layout_given.c_str(), &lc) != TAXOR_OK || taxor_hixf_set_layout(h, lc) != TAXOR_OK) die(taxor_gpu_last_error());
    }
    const uint64_t root_raw = taxor_hixf_ixf_raw_bytes(h, 0);
    std::vector<uint8_t> dz(3 * 16 * 64, 0);
    std::vector<int64_t> zero(64, 0), iota(64);
    for (uint64_t b = 0; b < 64; ++b) iota[b] = (int64_t)b;
    taxor_ixf_view dummy{64, 64, 16, 1, dz.data(), zero.data(), iota.data(), 0};
    taxor_hixf_view rv = *view;
    rv.n_ixf = 1;
    rv.ixf = &dummy;
    rv.n_user_bins = 64;
    rv.source = nullptr;
    rv.ixf_layout = 0;
    taxor_gpu_index *gi = nullptr;
    if (taxor_gpu_index_create(&rv, device, &gi) != TAXOR_OK) die(taxor_gpu_last_error());
    taxor_gpu_search_params prm{};
    prm.ratio = 0.05;
    prm.model = TAXOR_THR_PERCENTAGE;
    taxor_gpu_searcher *sr = nullptr;
    if (taxor_gpu_searcher_create(gi, &prm, &sr) != TAXOR_OK) die(taxor_gpu_last_error());
    auto hashes_of = [&](const std::vector<const PinRead *> &reads, std::vector<uint64_t> &hoff, std::vector<uint64_t> &hs) {
        std::string bases;
        std::vector<uint64_t> offs{0};
        for (const PinRead *r : reads) { bases += r->seq; offs.push_back(bases.size()); }
        const uint64_t *po = nullptr, *ph = nullptr;
        if (taxor_gpu_syncmers(sr, bases.data(), offs.data(), reads.size(), &po, &ph) != TAXOR_OK) die(taxor_gpu_last_error());
        hoff.assign(po, po + reads.size() + 1);
        hs.assign(ph, ph + hoff.back());
    };
    uint32_t arith = 0;
    std::string arith_source;
    taxor_ixf_variant chosen;
    {
        const taxor_ixf_view &root = view->ixf[0];
        taxor_ixf_variant_default(&chosen, root.seed, root.seg_len, root.src_stride ? root.src_stride : root.stride);
        chosen.layout = (uint16_t)view->ixf_layout;
    }
    if (!arith_given.empty()) {
        if (!parse_arith_spec(arith_given, &arith)) die("--ixf-arithmetic: expected kh=..,sm=..,rot=..,red=..,fp=..");
        taxor_ixf_arith_decode(arith, &chosen);
        arith_source = "given on the command line";
    } else {
        std::vector<const PinRead *> pos;
        for (const auto &s : scored)
            if (!wanted[s.id].seq.empty()) pos.push_back(&wanted[s.id]);
        if (pos.empty()) die("taxor pin: the expected TSV reports no classified read with >= 50 hashes that is also in the query file; name the reading with --ixf-arithmetic");
        std::vector<uint64_t> hoff, hs, s_off, s_hs;
        hashes_of(pos, hoff, hs);
        cap_hash_lists(hoff.data(), hs.data(), pos.size(), 160, s_off, s_hs);
        const taxor_ixf_view root = view->ixf[0];
        std::vector<taxor_ixf_variant> vs = variant_family(root, root_raw, view->ixf_layout);
        std::vector<float> ratio(vs.size() * pos.size());
        if (taxor_gpu_ixf_variant_scan(device, root.data, root_raw, root.bins, vs.data(), (uint32_t)vs.size(), s_hs.data(), s_off.data(), pos.size(), ratio.data()) != TAXOR_OK)
            die(taxor_gpu_last_error());
        const auto rank = rank_variants(vs, ratio, pos.size());
        printf("== variants: %zu readings of the root IXF's %llu raw bytes probed with %zu reads the expected TSV reports at match ratios %.3f .. %.3f\n", vs.size(),
               (unsigned long long)root_raw, pos.size(), scored.back().ratio, scored.front().ratio);
        char desc[512];
        for (size_t i = 0; i < std::min<size_t>(3, rank.size()); ++i) {
            taxor_ixf_variant_describe(&vs[rank[i].second], desc, sizeof desc);
            printf("  %.4f  %s\n", rank[i].first, desc);
        }
        if (rank.empty() || rank[0].first < 0.35f)
            die("taxor pin: no reading of the IXF answers for the reads the reference classified (best median match ratio " +
                std::to_string(rank.empty() ? 0.0 : rank[0].first) + "): the key hash or the record layout is not what this library assumes -- see `taxor probe`");
        chosen = vs[rank[0].second];
        // every IXF of the file under the layout that answered (strides, segment lengths, pitches recomputed from the array lengths)
        if (chosen.layout != view->ixf_layout && taxor_hixf_set_layout(h, chosen.layout) != TAXOR_OK) {
            // two pitch rules can give the ROOT the same pitch (64 k bins: padded == unpadded) while only one fits the other IXFs
            const std::string first_err = taxor_gpu_last_error();
            bool ok = false;
            for (uint32_t rule : {(uint32_t)taxor::IXF_PITCH_PADDED, (uint32_t)taxor::IXF_PITCH_BINS, (uint32_t)taxor::IXF_PITCH_STORED}) {
                const uint32_t alt = ((uint32_t)chosen.layout & ~(uint32_t)taxor::IXF_PITCH_MASK) | rule;
                if (alt == chosen.layout || taxor::ixf_layout_kind(alt) == taxor::IXF_KIND_BIT_SLICED) continue;
                if (taxor_hixf_set_layout(h, alt) == TAXOR_OK && (view->ixf[0].src_stride ? view->ixf[0].src_stride : view->ixf[0].stride) == chosen.stride) { chosen.layout = (uint16_t)alt; ok = true; break; }
            }
            if (!ok) die("taxor pin: " + first_err);
        }
        if (chosen.seg_len != view->ixf[0].seg_len || chosen.seed != view->ixf[0].seed)
            die("taxor pin: the answering reading differs in seed / segment length from what the loader took from the file; `taxor probe` shows the record layout");
        arith = taxor_ixf_arith_code(&chosen);
        arith_source = "variant scan of the root IXF over " + std::to_string(pos.size()) + " reads of the expected TSV (median best-bin match ratio " +
                       std::to_string(rank[0].first) + ")";
    }
    const std::string spec = arith_spec(chosen);
    const uint32_t layout = view->ixf_layout;
    char layout_desc[128];
    taxor_ixf_layout_describe(layout, layout_desc, sizeof layout_desc);
    bool relayout = taxor::ixf_layout_kind(layout) != taxor::IXF_KIND_ROWS || (layout & taxor::IXF_ROWS_POSITION_MAJOR);
    for (uint64_t i = 0; i < view->n_ixf && !relayout; ++i) relayout = view->ixf[i].src_stride != 0 && view->ixf[i].src_stride != view->ixf[i].stride;
    printf("== arithmetic: %s (code %u%s) -- %s\n", spec.c_str(), arith, arith == 0 ? ", this library's reading" : "", arith_source.c_str());
    printf("== layout: %s (code %u)%s\n", layout_desc, layout, relayout ? " -- transposed into the search layout on the device at load" : " -- the search layout, uploaded as it lies");

    // ---- 4a. the fixture reads: hashes, probe rows, fingerprint bytes (host mapping of the file) -----------------------------
    std::vector<const PinRead *> fx;
    for (const auto &id : fixture_ids)
        if (!wanted[id].seq.empty()) fx.push_back(&wanted[id]);
    std::vector<uint64_t> fx_hoff{0}, fx_hs;
    if (!fx.empty()) hashes_of(fx, fx_hoff, fx_hs);
    taxor_gpu_searcher_destroy(sr);
    taxor_gpu_index_destroy(gi);
    // parent of every IXF (ixf, merged bin) and the technical bins of every user bin
    const uint64_t n_ixf = view->n_ixf;
    std::vector<std::pair<int64_t, int64_t>> parent(n_ixf, {-1, -1});
    std::map<int64_t, std::vector<std::pair<uint64_t, uint64_t>>> bins_of;        // user bin -> (ixf, technical bin)
    for (uint64_t i = 0; i < n_ixf; ++i)
        for (uint64_t b = 0; b < view->ixf[i].bins; ++b) {
            const int64_t fn = view->ixf[i].fname_idx[b];
            if (fn < 0) {
                const int64_t ch = view->ixf[i].next_ixf[b];
                if (ch > 0 && (uint64_t)ch < n_ixf) parent[(size_t)ch] = {(int64_t)i, (int64_t)b};
            } else
                bins_of[fn].push_back({i, b});
        }
    std::map<std::string, int64_t> ub_of_accession;          // first species wins, like taxor_search.cpp:174
    for (uint64_t i = 0; i < meta->n_species; ++i)
        ub_of_accession.emplace(meta->species[i].accession_id ? meta->species[i].accession_id : "", (int64_t)meta->species[i].user_bin);

    // ---- 3. the search itself, as a child process, compared with the reference's TSV ----------------------------------------
    const std::string ours = out_file + ".tsv.tmp";
    {
        std::vector<std::string> av{"/proc/self/exe", "search", "--index-file", index_file, "--query-file", query_file, "--output-file", ours,
                                    "--threads", std::to_string(threads), "--error-rate", std::to_string(error_rate), "--gpu", std::to_string(device)};
        if (percentage >= 0.0) { av.push_back("--percentage"); av.push_back(std::to_string(percentage)); }
        if (arith != 0) { av.push_back("--ixf-arithmetic"); av.push_back(spec); }
        if (relayout || !layout_given.empty()) { av.push_back("--ixf-layout"); av.push_back(layout_desc); }
        std::vector<char *> cav;
        for (auto &s : av) cav.push_back(&s[0]);
        cav.push_back(nullptr);
        fflush(stdout);
        fflush(stderr);
        // a CHILD process (this one has used the GPU for the scan; the search gets a runtime of its own)
        pid_t pid = 0;
        if (posix_spawn(&pid, "/proc/self/exe", nullptr, nullptr, cav.data(), ::environ) != 0) die("taxor pin: cannot start the search process");
        int status = 0;
        if (waitpid(pid, &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0)
            die("taxor pin: the search of " + query_file + " failed (exit status " + std::to_string(WIFEXITED(status) ? WEXITSTATUS(status) : -1) + ")");
    }
    printf("== search vs the expected TSV\n");
    const uint64_t bad = compare_tsv(ours, expect_file);
    std::map<std::string, std::vector<std::string>> ours_lines;
    uint64_t n_ours = 0, n_same = 0;
    {
        FILE *f = fopen(ours.c_str(), "rb");
        if (!f) die("cannot open " + ours);
        std::string text;
        std::vector<char> buf(1 << 20);
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) text.append(buf.data(), n);
        fclose(f);
        size_t a = 0;
        while (a < text.size()) {
            size_t b = text.find('\n', a);
            if (b == std::string::npos) b = text.size();
            const std::string l = text.substr(a, b - a);
            a = b + 1;
            if (l.empty() || l[0] == '#') continue;
            ours_lines[l.substr(0, l.find('\t'))].push_back(l);
        }
        n_ours = ours_lines.size();
        for (const auto &kv : ours_lines) {
            const auto it = expect.find(kv.first);
            if (it != expect.end() && it->second == kv.second) ++n_same;
        }
    }
    remove(ours.c_str());

    // ---- 4b. write the fixture ---------------------------------------------------------------------------------------------------
    struct stat sb {};
    stat(index_file.c_str(), &sb);
    std::string js = "{\n";
    auto kv = [&](const std::string &k, const std::string &v, bool quote, bool last = false) {
        js += "  \"" + k + "\": " + (quote ? "\"" + json_escape(v) + "\"" : v) + (last ? "\n" : ",\n");
    };
    kv("format", "1", false);
    kv("name", name, true);
    kv("created_by", "taxor pin (taxor_amd/csrc/pin_cmd.h)", true);
    kv("pinned", bad == 0 ? "true" : "false", false);
    js += "  \"index\": {\"file\": \"" + json_escape(index_file.substr(index_file.find_last_of('/') == std::string::npos ? 0 : index_file.find_last_of('/') + 1)) +
          "\", \"bytes\": " + std::to_string((unsigned long long)sb.st_size) + ", \"k\": " + std::to_string(view->kmer_size) + ", \"s\": " +
          std::to_string(view->syncmer_size) + ", \"t\": " + std::to_string(view->t_syncmer) + ", \"scaling\": " + std::to_string(view->scaling) +
          ", \"window_size\": " + std::to_string((unsigned long long)view->window_size) + ", \"n_ixf\": " + std::to_string((unsigned long long)n_ixf) +
          ", \"n_user_bins\": " + std::to_string((unsigned long long)view->n_user_bins) + ", \"n_species\": " + std::to_string((unsigned long long)meta->n_species) +
          ", \"foreign_schema\": " + (meta->foreign_schema ? "true" : "false") + ",\n            \"schema\": {\"n_before\": " + std::to_string(schema.n_before) +
          ", \"n_after\": " + std::to_string(schema.n_after) + ", \"idx_bins\": " + std::to_string(schema.idx_bins) + ", \"idx_stride\": " + std::to_string(schema.idx_stride) +
          ", \"idx_seg_len\": " + std::to_string(schema.idx_seg_len) + ", \"idx_seed\": " + std::to_string(schema.idx_seed) + ", \"seg_len_is_rows\": " +
          std::to_string(schema.seg_len_is_rows) + ", \"default_seed\": " + std::to_string((unsigned long long)schema.default_seed) + ", \"len_unit\": " +
          std::to_string(schema.len_unit) + ", \"skip_before_len\": " + std::to_string(schema.skip_before_len) + ", \"skip_after_len\": " + std::to_string(schema.skip_after_len) + "},\n            \"ixf_arith\": " +
          std::to_string(arith) + ", \"arith_spec\": \"" + spec + "\", \"arith_source\": \"" + json_escape(arith_source) + "\",\n            \"ixf_layout\": " +
          std::to_string(layout) + ", \"layout_spec\": \"" + layout_desc + "\", \"relayout\": " + (relayout ? "true" : "false") + "},\n";
    js += "  \"search\": {\"error_rate\": " + std::to_string(error_rate) + ", \"percentage\": " + std::to_string(percentage) + "},\n";
    js += "  \"summary\": {\"reads_expected\": " + std::to_string(expect.size()) + ", \"reads_ours\": " + std::to_string((unsigned long long)n_ours) +
          ", \"identical\": " + std::to_string((unsigned long long)n_same) + ", \"differing_or_missing\": " + std::to_string((unsigned long long)bad) + "},\n";
    js += "  \"reads\": [\n";
    for (size_t r = 0; r < fx.size(); ++r) {
        const PinRead &pr = *fx[r];
        const uint64_t *hs = fx_hs.data() + fx_hoff[r];
        const size_t nh = (size_t)(fx_hoff[r + 1] - fx_hoff[r]);
        js += "    {\"id\": \"" + json_escape(pr.id) + "\", \"seq\": \"" + json_escape(pr.seq) + "\", \"n_hashes\": " + std::to_string(nh) +
              ", \"hashes_u64_b64\": \"" + b64(hs, nh * 8) + "\",\n     \"expect\": [";
        for (size_t i = 0; i < pr.expect.size(); ++i) js += std::string(i ? ", " : "") + "\"" + json_escape(pr.expect[i]) + "\"";
        js += "],\n     \"ours\": [";
        const auto &ol = ours_lines[pr.id];
        for (size_t i = 0; i < ol.size(); ++i) js += std::string(i ? ", " : "") + "\"" + json_escape(ol[i]) + "\"";
        js += "],\n     \"ixfs\": [";
        // the IXFs on the paths to the first three reported references: leaf bins of the user bin + the merged bins above them
        std::map<uint64_t, std::vector<std::pair<uint64_t, int>>> touched;       // ixf -> (technical bin, line index or -1 for a merged bin)
        size_t lines_done = 0;
        for (size_t li = 0; li < pr.expect.size() && lines_done < 3; ++li) {
            const auto f = split_tabs(pr.expect[li]);
            if (f.size() < 10 || f[1] == "-") continue;
            const auto ub = ub_of_accession.find(f[1]);
            if (ub == ub_of_accession.end()) continue;
            ++lines_done;
            for (const auto &ib : bins_of[ub->second]) {
                touched[ib.first].push_back({ib.second, (int)li});
                for (int64_t v = (int64_t)ib.first; v > 0 && parent[(size_t)v].first >= 0; v = parent[(size_t)v].first) {
                    auto &tv = touched[(uint64_t)parent[(size_t)v].first];
                    const std::pair<uint64_t, int> e{(uint64_t)parent[(size_t)v].second, -1};
                    if (std::find(tv.begin(), tv.end(), e) == tv.end()) tv.push_back(e);
                }
            }
        }
        bool first_ixf = true;
        for (const auto &tk : touched) {
            const taxor_ixf_view &X = view->ixf[tk.first];
            std::vector<uint32_t> rows(3 * nh);
            std::vector<uint8_t> fps(nh);
            for (size_t i = 0; i < nh; ++i) {
                const taxor::ixf_probe p = taxor::ixf_probe_key_arith(hs[i], X.seed, (uint32_t)X.seg_len, arith);
                rows[3 * i] = p.row[0]; rows[3 * i + 1] = p.row[1]; rows[3 * i + 2] = p.row[2];
                fps[i] = (uint8_t)(p.fp4 & 0xFFu);
            }
            js += std::string(first_ixf ? "\n" : ",\n") + "       {\"ixf\": " + std::to_string((unsigned long long)tk.first) + ", \"bins\": " + std::to_string((unsigned long long)X.bins) +
                  ", \"stride\": " + std::to_string((unsigned long long)X.stride) + ", \"seg_len\": " + std::to_string((unsigned long long)X.seg_len) + ", \"seed\": " +
                  std::to_string((unsigned long long)X.seed) + ", \"rows_u32_b64\": \"" + b64(rows.data(), rows.size() * 4) + "\", \"fingerprints_u8_b64\": \"" +
                  b64(fps.data(), fps.size()) + "\",\n        \"bins_probed\": [";
            first_ixf = false;
            for (size_t bi = 0; bi < tk.second.size(); ++bi) {
                const uint64_t bin = tk.second[bi].first;
                std::vector<uint8_t> bytes(3 * nh);
                uint32_t count = 0;
                for (size_t i = 0; i < nh; ++i) {
                    for (int j = 0; j < 3; ++j)          // read where the FILE keeps it (ixf_layout.h)
                        bytes[3 * i + j] = taxor::ixf_src_fingerprint(X.data, layout, rows[3 * i + j], bin, X.seg_len, X.src_stride ? X.src_stride : X.stride, X.bins);
                    count += (uint8_t)(bytes[3 * i] ^ bytes[3 * i + 1] ^ bytes[3 * i + 2]) == fps[i];
                }
                js += std::string(bi ? ", " : "") + "{\"bin\": " + std::to_string((unsigned long long)bin) + ", \"expect_line\": " + std::to_string(tk.second[bi].second) +
                      ", \"merged\": " + (tk.second[bi].second < 0 ? "true" : "false") + ", \"count\": " + std::to_string(count) + ", \"bytes_u8_b64\": \"" +
                      b64(bytes.data(), bytes.size()) + "\"}";
            }
            js += "]}";
        }
        js += "]}";
        js += r + 1 < fx.size() ? ",\n" : "\n";
    }
    js += "  ]\n}\n";
    {
        FILE *f = fopen(out_file.c_str(), "wb");
        if (!f) die("cannot write " + out_file);
        fwrite(js.data(), 1, js.size(), f);
        fclose(f);
    }
    printf("== fixture: %s (%zu reads with hashes, probe rows and fingerprint bytes; %.1f KB)\n", out_file.c_str(), fx.size(), js.size() / 1024.0);
    printf("%s\n", bad == 0 ? "PINNED: every read's output equals the reference's; commit the fixture and run `pytest tests/test_real_fixtures.py`"
                            : "NOT PINNED: the output differs from the reference's (fixture written with \"pinned\": false, for diagnosis)");
    taxor_hixf_free(h);
    return bad == 0 ? 0 : 3;
}
