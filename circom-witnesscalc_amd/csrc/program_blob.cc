// Structural validation of a program that did not come out of compile_program, and the pointer-free blob that travels between
// processes and GPUs (program.hpp).
#include "compile_internal.hpp"

namespace cwc {

// ---- structural validation of a program that did not come out of compile_program (an imported blob) ----------------
// Everything the interpreter and the pack kernel address through the program is checked against the tile and LDS
// geometry: a truncated or corrupted broadcast must fail here, not read or write out of bounds on the device.
bool validate_program(const Program& p, std::string& err) {
    auto bad = [&](const std::string& m) {
        err = "invalid program: " + m;
        return false;
    };
    const uint32_t T = p.T, G = p.G;
    if (T == 0 || T > 64 || (T & (T - 1)) || G != 64 / T) return bad("tile geometry");
    if (p.divider != 0 && p.divider != 1 && p.divider != 3 && p.divider != 4) return bad("divider mode");
    if (p.divider && T == 64) return bad("divider program at tile width 64");
    const uint64_t tile_bytes = ws_tile_bytes(p.n_const, p.n_slots, T);
    if (p.n_const == 0 || p.n_slots == 0 || tile_bytes > 0xffffffffull) return bad("tile size");
    if ((uint64_t)p.n_bundles * G * 16ull > 0xffffffffull) return bad("record stream size");
    if (p.hdr.size() != p.n_bundles || p.recs.size() != (size_t)p.n_bundles * G * 4 || p.crefs.size() != (size_t)p.n_cref_rows * G ||
        p.consts.size() != (size_t)p.n_const * 8 || p.witness_refs.size() != p.n_witness || p.div_lanes.size() != p.n_div_requests)
        return bad("array sizes");
    const uint32_t slot_bytes = 32u * T, HI = 16u * T;
    const uint64_t trash_slot = ((uint64_t)p.n_const + p.n_slots) * slot_bytes;
    if (p.trash_off != OFF_NOWHERE && p.trash_off != trash_slot) return bad("trash offset");
    if (tile_bytes >= (uint64_t)OFF_NOWHERE) return bad("tile size");
    const uint64_t trash_off = p.trash_off;
    // streams: consecutive bundle ranges, each starting at a multiple of the pipeline depths; one stream unless the
    // divider mode is none or one divider wave per interpreter
    const uint32_t NS = p.n_streams;
    if (NS != 1 && NS != 2 && NS != 4) return bad("stream count");
    if (NS > 1 && p.divider > 1) return bad("streams with a shared divider wave");
    uint32_t next_first = 0, req_sum = 0;
    for (uint32_t s = 0; s < NS; ++s) {
        if (p.stream_first[s] < next_first || (p.stream_first[s] % 4) != 0 || (uint64_t)p.stream_first[s] + p.stream_count[s] > p.n_bundles) return bad("stream ranges");
        next_first = p.stream_first[s] + p.stream_count[s];
        req_sum += p.stream_div_requests[s];
    }
    if (p.stream_first[0] != 0 || (NS == 1 && p.stream_count[0] != p.n_bundles) || req_sum != p.n_div_requests) return bad("stream ranges");
    uint32_t n_req = 0, n_get = 0, n_posts = 0;
    bool in_flight = false;
    uint32_t stream = 0, stream_req = 0, cref_row = 0;
    for (uint32_t b = 0; b < p.n_bundles; ++b) {
        while (stream + 1 < NS && b >= p.stream_first[stream + 1]) {
            if (in_flight || stream_req != p.stream_div_requests[stream]) return bad("division requests of stream " + std::to_string(stream));
            ++stream;
            stream_req = 0;
        }
        // (checked for the stream b belongs to: the interpreter wave of stream s starts its row counter at stream_cref_first[s])
        if (b == p.stream_first[stream] && p.stream_count[stream] && p.stream_cref_first[stream] != cref_row) return bad("third-operand rows of stream " + std::to_string(stream));
        const bool executed = b < p.stream_first[stream] + p.stream_count[stream];
        const uint32_t h = p.hdr[b], cls = h & HDR_CLASS_MASK, cnt = (h >> HDR_COUNT_SHIFT) & 0x7f;
        if (cls >= C_COUNT || (cls != C_SCAN && (h >> 19) != 0)) return bad("bundle " + std::to_string(b) + ": header");
        if (cls == C_SCAN) {  // pairs of record positions, an iteration count that covers the longest chain segment, a shift below 254
            const uint32_t iters = (h >> HDR_SCAN_ITER_SHIFT) + 1u, sh = (h >> HDR_SCAN_SHIFT_SHIFT) & 0xffu;
            if (h & HDR_SCAN_CONV) {  // the 2k - 1 columns of a k x k limb product
                if (T > SCAN_MAX_T || cnt != 2 * iters - 1 || iters < 2 || sh != 0 || (h & 0x7e800u)) return bad("bundle " + std::to_string(b) + ": convolution bundle");
            } else {
                // one kind per bundle: carry chain, long division by one limb, borrow chain, comparison (the last with its two result bits; its shift field: 1 = the chain's bits in Montgomery form)
                const uint32_t kinds = h & (HDR_SCAN_DIV | HDR_SCAN_BORROW | HDR_SCAN_LEX);
                const bool sel = kinds == (HDR_SCAN_BORROW | HDR_SCAN_LEX);  // a bundle of selections: the comparison's code in the shift field
                const bool kind_ok = sel ? (!(h & (HDR_SCAN_KG | HDR_SCAN_KL)) && (sh & 7u) >= SEL_LT && (sh & 7u) <= SEL_NEZ && sh < 16u && iters == 1u)
                                         : ((kinds & (kinds - 1u)) == 0 && (!(h & (HDR_SCAN_KG | HDR_SCAN_KL)) || (h & HDR_SCAN_LEX)) && (!(h & HDR_SCAN_LEX) || sh <= 1u));
                if (T > SCAN_MAX_T || (cnt & 1u) || cnt == 0 || iters > cnt / 2 || sh >= 254u || (h & 0x61000u) || !kind_ok) return bad("bundle " + std::to_string(b) + ": scan bundle");
            }
        }
        // posts and waits are C_SYNC bundles without nodes: stream 0 posts once, every other stream waits in its first bundle
        // (nothing else is compiled).  (Bits 11-18 of a scan bundle's header are its own -- kind, result bits --, checked above.)
        const uint32_t hg = cls == C_SCAN ? (h & ~0x7f800u) : h;
        if (((hg & (HDR_POST | HDR_WAIT)) != 0) != (cls == C_SYNC) || (cls == C_SYNC && cnt != 0)) return bad("bundle " + std::to_string(b) + ": post / wait bits");
        if ((hg & HDR_POST) && !(NS > 1 && stream == 0 && executed && n_posts++ == 0)) return bad("bundle " + std::to_string(b) + ": post");
        if (((hg & HDR_WAIT) != 0) != (NS > 1 && stream != 0 && executed && b == p.stream_first[stream])) return bad("bundle " + std::to_string(b) + ": wait");
        // the staging loads of the two bundles behind a wait are issued in front of it: they must not read anything
        if (NS > 1 && stream != 0 && executed && (b == p.stream_first[stream] + 1 || b == p.stream_first[stream] + 2) && cnt != 0) return bad("bundle " + std::to_string(b) + ": work right behind a wait");
        if ((hg & (HDR_A_CANON | HDR_B_CANON)) && !(cls == C_BIT || cls == C_IDIVMOD || cls == C_CMPS)) return bad("bundle " + std::to_string(b) + ": operand form bits");
        if ((hg & HDR_OUT_CANON) && !(cls == C_BIT || cls == C_IDIVMOD || cls == C_CMPS || cls == C_CMPZ || cls == C_INPUT)) return bad("bundle " + std::to_string(b) + ": result form bit");
        const uint32_t rep = cls == C_MULQ || cls == C_MULF ? COOP_LANES : 1u;
        if ((cnt == 0 && cls != C_LIN && cls != C_SYNC) || cnt * rep > G) return bad("bundle " + std::to_string(b) + ": node count");
        if (cls == C_MULF && T > COOP_FUSE_MAX_T) return bad("bundle " + std::to_string(b) + ": fused bundle at this tile width");
        if (!executed && cnt != 0) return bad("bundle " + std::to_string(b) + ": outside every stream");
        if (cls == C_MULQ && T > COOP_MAX_T) return bad("bundle " + std::to_string(b) + ": narrow bundle at this tile width");
        if ((cls == C_DIVREQ || cls == C_DIVGET) && !p.divider) return bad("bundle " + std::to_string(b) + ": request / collect without a divider");
        if (cls == C_DIV && p.divider) return bad("bundle " + std::to_string(b) + ": inline division in a divider program");
        if (cls == C_DIVREQ) {
            if (in_flight || n_req >= p.n_div_requests || cnt * T > mbox_lanes(p.divider) || p.div_lanes[n_req] != cnt * T) return bad("bundle " + std::to_string(b) + ": division request");
            in_flight = true;
            ++n_req;
            ++stream_req;
        }
        if (cls == C_DIVGET) {
            if (!in_flight) return bad("bundle " + std::to_string(b) + ": collect without a request");
            in_flight = false;
            ++n_get;
        }
        for (uint32_t q = 0; q < G; ++q) {
            const uint32_t* r = &p.recs[((size_t)b * G + q) * 4];
            // staging loads: 16 bytes per lane at off + 16 t and at off + HI + 16 t
            for (int k = 0; k < 2; ++k)
                if (r[k] != OFF_NOWHERE && ((r[k] % slot_bytes) != 0 || (uint64_t)r[k] + slot_bytes > tile_bytes)) return bad("bundle " + std::to_string(b) + ": operand offset");
            if (cls == C_MULF) {  // stage codes: op2 in the main records (even positions), op3 (additions only) in the extra records
                const uint32_t code = r[2] & CTRL_SUB_MASK;
                if ((q & 1u) ? (code == FOP_MUL || code > FOP_RSUB || (r[2] & ~CTRL_MASK) != trash_off) : code > FOP_RSUB) return bad("bundle " + std::to_string(b) + ": fused stage code");
                if ((code == FOP_MUL && !(h & HDR_F_S2MUL)) || (code > FOP_MUL && !(h & ((q & 1u) ? HDR_F_S3LIN : HDR_F_S2LIN)))) return bad("bundle " + std::to_string(b) + ": fused stage bits");
            }
            if (cls == C_SCAN && (h & HDR_SCAN_CONV)) {
                if (q < cnt && ((r[2] & CTRL_SUB_MASK) || !(r[2] & CTRL_ACTIVE))) return bad("bundle " + std::to_string(b) + ": convolution record");
            } else if (cls == C_SCAN && q < cnt) {  // position 2p: the step's OUT record, 2p + 1: its ACC record, same START bit; the first pair starts a chain
                const uint32_t sub = r[2] & CTRL_SUB_MASK, sub0 = p.recs[((size_t)b * G + (q & ~1u)) * 4 + 2] & CTRL_SUB_MASK;
                if ((sub & SCAN_ROLE_ACC) != (q & 1u) || (sub & ~(SCAN_ROLE_ACC | SCAN_START)) || ((sub ^ sub0) & SCAN_START) || (q < 2 && !(sub & SCAN_START)) || !(r[2] & CTRL_ACTIVE))
                    return bad("bundle " + std::to_string(b) + ": scan record");
            }
            const uint32_t dst = r[2] & ~CTRL_MASK;
            if (dst != trash_off && ((dst % slot_bytes) != 0 || dst < (uint64_t)p.n_const * slot_bytes || dst >= trash_slot)) return bad("bundle " + std::to_string(b) + ": destination");
            const uint32_t la = r[3] & 0xffffu, lb = r[3] >> 16;
            const bool bitx = cls == C_BIT && (r[2] & CTRL_SUB_MASK) == SUB_BITX;
            if ((la % 16) != 0 || la + 16u * (T - 1) + LDS_HALF_BYTES + 16u > LDS_BYTES) return bad("bundle " + std::to_string(b) + ": LDS address");
            // (a bit-extract lane carries its shift amount there; idle lanes of such a bundle keep a stage address, unused)
            const bool active = (r[2] & CTRL_ACTIVE) != 0;
            if (bitx ? (active && lb / 16 >= 254) : ((lb % 16) != 0 || lb + 16u * (T - 1) + LDS_HALF_BYTES + 16u > LDS_BYTES)) return bad("bundle " + std::to_string(b) + ": LDS address");
            const bool has_row = cls == C_INPUT || cls == C_TERN;
            if (has_row && cref_row >= p.n_cref_rows) return bad("bundle " + std::to_string(b) + ": third-operand row");
            const uint32_t cr = has_row ? p.crefs[(size_t)cref_row * G + q] : 0u;
            if (cls == C_INPUT && cr >= p.n_inputs) return bad("bundle " + std::to_string(b) + ": input index");
            if (cls == C_TERN && ((cr % slot_bytes) != 0 || (uint64_t)cr + slot_bytes > tile_bytes)) return bad("bundle " + std::to_string(b) + ": third operand");
        }
        (void)HI;
        cref_row += cls == C_INPUT || cls == C_TERN;
    }
    if (cref_row != p.n_cref_rows) return bad("third-operand rows");
    bool any_fused = false, any_scan = false;  // (one interpreter instance each: a program has one kind or the other)
    for (uint32_t h : p.hdr) {
        any_fused = any_fused || (h & HDR_CLASS_MASK) == C_MULF;
        any_scan = any_scan || (h & HDR_CLASS_MASK) == C_SCAN || ((h & HDR_CLASS_MASK) == C_MUL && (h & HDR_MUL_CC));
        if ((h & HDR_CLASS_MASK) == C_MUL && (h & HDR_MUL_CC) && (T > SCAN_MAX_T || (h & (HDR_LIN_ADD | HDR_LIN_SUB)))) return bad("canonical-product bundle");
    }
    if (any_fused && any_scan) return bad("fused and scan / canonical-product bundles in one program");
    if ((any_scan || any_fused) && p.divider > 1) return bad("scan / fused bundles in a program for these divider waves");  // (kernels.hip launch_interp: which instances exist)
    if (in_flight || n_req != p.n_div_requests || n_get != n_req || stream_req != p.stream_div_requests[stream]) return bad("division requests");
    if (NS > 1 && n_posts != 1) return bad("streams without a post");
    for (uint32_t w : p.witness_refs)
        if ((w & REF_CONST) ? (w & ~REF_CONST) >= p.n_const : (w & ~REF_CANON) >= p.n_slots) return bad("witness reference");
    return true;
}

// ---- blob ----------------------------------------------------------------------------------------
static const uint32_t kBlobMagic = 0x47505743u;  // "CWPG"
struct BlobHeader {
    uint32_t magic, version, T, G, n_bundles, n_slots, n_const, n_inputs, n_witness, divider, n_div_requests, n_streams;
    uint32_t stream_first[MAX_STREAMS], stream_count[MAX_STREAMS], stream_div_requests[MAX_STREAMS], stream_cref_first[MAX_STREAMS];
    uint32_t n_cref_rows, trash_off;
    double stream_cycles[MAX_STREAMS], stream_cycles_mul_div[MAX_STREAMS], stream_chain_cycles[MAX_STREAMS];
    ProgramStats stats;
};

size_t program_blob_size(const Program& p) {
    return sizeof(BlobHeader) + 4 * (p.hdr.size() + p.recs.size() + p.crefs.size() + p.consts.size() + p.witness_refs.size() + p.div_lanes.size());
}

// the blob at dst (program_blob_size(p) bytes): what gwb_graph_export writes straight into the caller's buffer
void program_blob_write(const Program& p, uint8_t* dst) {
    BlobHeader h;
    memset(&h, 0, sizeof h);
    h.magic = kBlobMagic;
    h.version = 20;  // (20: selection bundles (C_SCAN with both HDR_SCAN_BORROW and HDR_SCAN_LEX).  19: scan bundles of one-bit recurrences (HDR_SCAN_BORROW / HDR_SCAN_LEX).  18: convolution bundles (C_SCAN with HDR_SCAN_CONV), one more statistics word.  17: results without a slot go nowhere (OFF_NOWHERE) instead of a trash slot.  16: scan bundles, class 14, in place of round 3's macro bundles; one more statistics word.  15: blob_checksum in the image's trailer)
    h.T = p.T; h.G = p.G; h.n_bundles = p.n_bundles; h.n_slots = p.n_slots; h.n_const = p.n_const;
    h.n_inputs = p.n_inputs; h.n_witness = p.n_witness;
    h.divider = p.divider; h.n_div_requests = p.n_div_requests;
    h.n_streams = p.n_streams;
    h.n_cref_rows = p.n_cref_rows;
    h.trash_off = p.trash_off;
    for (uint32_t s = 0; s < MAX_STREAMS; ++s) {
        h.stream_first[s] = p.stream_first[s]; h.stream_count[s] = p.stream_count[s]; h.stream_div_requests[s] = p.stream_div_requests[s]; h.stream_cref_first[s] = p.stream_cref_first[s];
        h.stream_cycles[s] = p.stream_cycles[s]; h.stream_cycles_mul_div[s] = p.stream_cycles_mul_div[s]; h.stream_chain_cycles[s] = p.stream_chain_cycles[s];
    }
    h.stats = p.stats;
    memcpy(dst, &h, sizeof h);
    dst += sizeof h;
    auto put = [&](const std::vector<uint32_t>& v) {
        if (!v.empty()) memcpy(dst, v.data(), 4 * v.size());
        dst += 4 * v.size();
    };
    put(p.hdr); put(p.recs); put(p.crefs); put(p.consts); put(p.witness_refs); put(p.div_lanes);
}

std::vector<uint8_t> program_to_blob(const Program& p) {
    std::vector<uint8_t> out(program_blob_size(p));
    program_blob_write(p, out.data());
    return out;
}

bool program_from_blob(const uint8_t* data, size_t len, Program& p, std::string& err) {
    BlobHeader h;
    if (len < sizeof h) { err = "program blob too short"; return false; }
    memcpy(&h, data, sizeof h);
    if (h.magic != kBlobMagic || h.version != 20 || h.T == 0 || h.T > 64 || h.G != 64 / h.T || (h.divider != 0 && h.divider != 1 && h.divider != 3 && h.divider != 4)) { err = "bad program blob header"; return false; }
    p = Program();
    p.T = h.T; p.G = h.G; p.n_bundles = h.n_bundles; p.n_slots = h.n_slots; p.n_const = h.n_const;
    p.n_inputs = h.n_inputs; p.n_witness = h.n_witness; p.stats = h.stats;
    p.divider = h.divider; p.n_div_requests = h.n_div_requests;
    p.n_streams = h.n_streams;
    p.n_cref_rows = h.n_cref_rows;
    p.trash_off = h.trash_off;
    for (uint32_t s = 0; s < MAX_STREAMS; ++s) {
        p.stream_first[s] = h.stream_first[s]; p.stream_count[s] = h.stream_count[s]; p.stream_div_requests[s] = h.stream_div_requests[s]; p.stream_cref_first[s] = h.stream_cref_first[s];
        p.stream_cycles[s] = h.stream_cycles[s]; p.stream_cycles_mul_div[s] = h.stream_cycles_mul_div[s]; p.stream_chain_cycles[s] = h.stream_chain_cycles[s];
    }
    const size_t n_hdr = p.n_bundles, n_recs = (size_t)p.n_bundles * p.G * 4, n_c = (size_t)p.n_cref_rows * p.G,
                 n_k = (size_t)p.n_const * 8, n_w = p.n_witness, n_d = p.n_div_requests;
    if (len != sizeof h + 4 * (n_hdr + n_recs + n_c + n_k + n_w + n_d)) { err = "program blob size mismatch"; return false; }
    const uint32_t* q = (const uint32_t*)(data + sizeof h);
    p.hdr.assign(q, q + n_hdr); q += n_hdr;
    p.recs.assign(q, q + n_recs); q += n_recs;
    p.crefs.assign(q, q + n_c); q += n_c;
    p.consts.assign(q, q + n_k); q += n_k;
    p.witness_refs.assign(q, q + n_w); q += n_w;
    p.div_lanes.assign(q, q + n_d);
    return true;
}

}  // namespace cwc
