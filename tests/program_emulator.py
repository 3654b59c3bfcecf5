"""Test-only emulator of a compiled program blob (gwb_graph_export) in Python big ints.

Validates the HOST side of the product -- level scheduling into bundles, liveness slot reuse, operand
encoding, witness references -- on machines without a GPU.  Arithmetic comes from oracle/model.py, so
this checks the compiler, not the HIP kernels (those are checked by the `-m gpu` parity tests)."""
import struct

from oracle import model

REF_CONST = 0x80000000
REF_CANON = 0x40000000  # the slot holds the canonical integer, not the Montgomery form
HDR_B_CANON, HDR_A_CANON, HDR_OUT_CANON = 1 << 14, 1 << 17, 1 << 18
CTRL_SUB_MASK, CTRL_ACTIVE, CTRL_MASK = 7, 8, 15
RING_BUNDLES, OPND_AHEAD, REC_AHEAD = 4, 2, 4
RING_SLOT_BYTES, LDS_HALF_BYTES, STAGE_BYTES = 2048, 1024, 4096
LDS_RING_OFF = 0
LDS_STAGE_OFF = LDS_RING_OFF + RING_BUNDLES * RING_SLOT_BYTES
LDS_REC_OFF = LDS_STAGE_OFF + OPND_AHEAD * STAGE_BYTES
SUB_NAMES = {"LIN": ["Add", "Sub"], "CMPZ": ["Eq", "Neq", "Land", "Lor"], "CMPS": ["Lt", "Gt", "Leq", "Geq"],
             "BIT": ["Shl", "Shr", "Bor", "Band", "Bxor", "BitX"], "IDIVMOD": ["Idiv", "Mod"], "MUL": ["Add", "Sub", "Mul"], "DIV": ["Div"],
             "MULQ": ["Add", "Sub", "Mul"]}
R_MONT = (1 << 256) % model.M
R_INV = pow(R_MONT, -1, model.M)
HDR_FMT = "<12I12I6I12d52Q"
HDR_POST, HDR_WAIT = 1 << 15, 1 << 16
HDR_SIZE = struct.calcsize(HDR_FMT)
CLASS_NAMES = ["INPUT", "MUL", "LIN", "DIV", "CMPZ", "CMPS", "BIT", "IDIVMOD", "TERN", "DIVREQ", "DIVGET", "MULQ", "SYNC", "MULF", "SCAN"]
N_CLASSES = len(CLASS_NAMES)
FOP_NONE, FOP_MUL, FOP_ADD, FOP_SUB, FOP_RSUB = range(5)  # stage codes of a fused node (class MULF)
HDR_F_S2MUL, HDR_F_S2LIN, HDR_F_S3LIN = 1 << 11, 1 << 12, 1 << 13
COOP_FUSE_MAX_T = 2
COOP_LANES, COOP_MAX_T = 4, 4
# scan bundles (class SCAN): pairs of record positions (2p: the step's OUT record, 2p + 1: its ACC record); header bit 11
# kind (0 carry chain, 1 long division by one limb), bits 19-26 the shift, bits 27-31 iterations - 1; sub-op bit 0 role, bit 1 START
HDR_MUL_CC = 1 << 13
OFF_NOWHERE = 0xFFFF0000
SCAN_MAX_T, HDR_SCAN_DIV, HDR_SCAN_CONV, HDR_SCAN_SHIFT_SHIFT, HDR_SCAN_ITER_SHIFT, SCAN_ROLE_ACC, SCAN_START = 2, 1 << 11, 1 << 12, 19, 27, 1, 2
# one-bit recurrences (round 5): borrow chain of a register-wise subtraction, most-significant-difference comparison with its two result bits
HDR_SCAN_BORROW, HDR_SCAN_LEX, HDR_SCAN_KG, HDR_SCAN_KL = 1 << 13, 1 << 14, 1 << 15, 1 << 16


def blob_checksum(body):
    """The trailer checksum of an exported image (bcast.cc blob_checksum): position-dependent sum over 64-bit words."""
    import numpy as np
    K1, K2, K3, M = 0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, (1 << 64) - 1
    n = len(body)
    w = np.frombuffer(bytes(body) + b"\0" * (-n % 8), dtype="<u8")
    with np.errstate(over="ignore"):
        x = (w ^ (np.arange(len(w), dtype=np.uint64) * np.uint64(K1))) * np.uint64(K2)
        x ^= x >> np.uint64(29)
        h = (int(x.sum(dtype=np.uint64)) + n * K3) & M
    h ^= h >> 32
    h = (h * K1) & M
    return h ^ (h >> 29)


class Blob:
    def __init__(self, data):
        h = struct.unpack_from(HDR_FMT, data, 0)
        (self.magic, self.version, self.T, self.G, self.n_bundles, self.n_slots, self.n_const, self.n_inputs,
         self.n_witness, self.divider, self.n_div_requests, self.n_streams) = h[:12]
        self.stream_first, self.stream_count, self.stream_div_requests, self.stream_cref_first = h[12:16], h[16:20], h[20:24], h[24:28]
        self.n_cref_rows, self.trash_off = h[28], h[29]
        self.stream_cycles = h[30:34]
        assert self.n_streams in (1, 2, 3, 4) and all(f % 4 == 0 for f in self.stream_first[:self.n_streams])
        assert sum(self.stream_div_requests[:self.n_streams]) == self.n_div_requests
        assert self.n_streams == 1 or self.divider in (0, 1), "streams have a divider wave each, or none"
        st = h[42:]
        c0, c1, c2 = 6, 6 + N_CLASSES, 6 + 2 * N_CLASSES
        self.stats = dict(n_nodes=st[0], n_op=st[1], n_input_nodes=st[2], n_const=st[3], n_witness=st[4], depth=st[5],
                          class_nodes=st[c0:c1], class_bundles=st[c1:c2], n_op_compiled=st[c2], n_bitx_bundles=st[c2 + 1], n_bitx_nodes=st[c2 + 2], algorithmic_bytes_per_set=st[c2 + 3], n_coop_rider_bundles=st[c2 + 4], n_conversions=st[c2 + 5], n_canonical=st[c2 + 6], form_cycles_saved=st[c2 + 7], n_folded=st[c2 + 8], n_numbered=st[c2 + 9], n_shaken=st[c2 + 10], n_fused_nodes=st[c2 + 11], n_scan_steps=st[c2 + 12], chain_floor_cycles=st[c2 + 13], n_conv_products=st[c2 + 14], depth_scan=st[c2 + 15])
        assert self.magic == 0x47505743 and self.G == 64 // self.T
        pos = HDR_SIZE

        def take(n):
            nonlocal pos
            v = struct.unpack_from("<%dI" % n, data, pos)
            pos += 4 * n
            return v
        self.hdr = take(self.n_bundles)
        self.recs = take(self.n_bundles * self.G * 4)
        self.crefs = take(self.n_cref_rows * self.G)  # one row per INPUT / TERN bundle, in bundle order
        k = take(self.n_const * 8)
        self.consts_raw = [sum(k[8 * i + j] << (32 * j) for j in range(8)) for i in range(self.n_const)]
        self.consts = [v * R_INV % model.M for v in self.consts_raw]
        assert self.n_const >= 1 and self.consts[-1] == 0  # trailing dummy entry (prefetch target)
        self.witness_refs = take(self.n_witness)
        self.div_lanes = take(self.n_div_requests)
        assert self.divider in (0, 1, 3, 4) and pos <= len(data)  # (an exported blob carries the input map behind the program)


def run(blob: Blob, inputs_row):
    """Evaluate one input set (list of ints) through the format-v4 program the way the interpreter kernel does, for
    t = 0; returns (witness ints, status bits).  Values are held the way the kernel holds them -- the Montgomery form
    x * 2^256 mod r, or the canonical integer where the compiler's representation inference keeps one -- and every
    operation is done on those words: a value read in the wrong form gives a wrong witness.  Models the timing rules of the pipeline: the staging load of bundle b
    sees the tile as it is after the stores of bundle b - OPND_AHEAD - 1 (it is issued before bundle b - OPND_AHEAD
    stores), a ring cell holds the result of the last bundle that wrote it.  Raises on any read of a slot or ring cell
    that does not hold the value the compiler meant (scheduling / liveness / encoding bug)."""
    T, G = blob.T, blob.G
    slot_bytes = 32 * T
    NC = blob.n_const
    # results without a slot and the operands of ring-forwarded / idle node slots: OFF_NOWHERE (dropped / zeros through the buffer
    # range check), or -- programs compiled with CWC_NOWHERE=0 -- the tile's trash slot and the zero constant's slot
    trash = blob.trash_off
    assert trash in (OFF_NOWHERE, (NC + blob.n_slots) * slot_bytes)
    zero_off = OFF_NOWHERE if trash == OFF_NOWHERE else (NC - 1) * slot_bytes
    history = {}  # value slot -> list of (stream, bundle that stored, value)
    status = 0
    # Streams (wavefronts of the tile with their own bundle ranges) run at their own pace; what orders them is stream
    # 0's post bundle (every result of its earlier bundles is in memory) and the wait bundle that every other stream
    # starts with (the loads issued from the next iteration on see those results: the staging loads of the bundle three
    # further on, the third-operand loads of the next bundle).  A slot belongs to the stream that writes it; another stream may read it only if it was written
    # once, before the post, and the read is issued behind the wait.
    post_at = None
    for b in range(blob.stream_first[0], blob.stream_first[0] + blob.stream_count[0]):
        if blob.hdr[b] & HDR_POST and CLASS_NAMES[blob.hdr[b] & 0xF] != "SCAN":  # (bits 11-18 of a scan bundle's header are its own)
            assert post_at is None
            post_at = b
    assert (post_at is not None) == (blob.n_streams > 1)

    def mem_at(off, as_of_bundle, stream, issued_at):
        """content of the tile at byte offset off, after the stores of the stream's bundles <= as_of_bundle; the load
        is issued at the top of bundle `issued_at`"""
        if off == OFF_NOWHERE:  # beyond the tile: the buffer range check lands zeros in the stage cell (idle node slots, the 0 of Neg = 0 - a)
            return 0
        assert off % slot_bytes == 0
        s_ = off // slot_bytes
        if s_ < NC:
            return blob.consts_raw[s_]
        assert s_ < NC + blob.n_slots, "operand read from the trash slot"
        hist = history[s_ - NC]
        if hist[0][0] != stream:
            # (the slot may have held earlier values of stream 0; its last store is the one every other stream sees)
            assert hist[0][0] == 0 and stream != 0, "only values of stream 0 cross streams"
            assert hist[-1][1] <= post_at - 1 and issued_at >= blob.stream_first[stream] + 1, "cross-stream read outside the post / wait order"
            cross_reads.add(s_ - NC)
            return hist[-1][2]
        for _, wb, val in reversed(hist):
            if wb <= as_of_bundle:
                return val
        raise AssertionError("slot read before it was written")

    n_requests_total = 0
    cross_reads = set()  # slots of stream 0 that other streams read: never written again behind the post
    bundles = [(s_, b) for s_ in range(blob.n_streams) for b in range(blob.stream_first[s_], blob.stream_first[s_] + blob.stream_count[s_])]
    assert blob.stream_first[0] == 0 and all(blob.stream_first[k] >= blob.stream_first[k - 1] + blob.stream_count[k - 1] for k in range(1, blob.n_streams))
    assert blob.stream_first[blob.n_streams - 1] + blob.stream_count[blob.n_streams - 1] == blob.n_bundles
    for stream, b in bundles:
        if b == blob.stream_first[stream]:
            ring = {}     # (ring cell, node slot) -> (bundle that wrote it, value)
            mailbox = None  # operands of the division request in flight (asynchronous divider programs)
            n_requests = 0
            cref_row = blob.stream_cref_first[stream]
        h = blob.hdr[b]
        cls, cnt = h & 0xF, (h >> 4) & 0x7F
        name = CLASS_NAMES[cls]
        h_scan = h
        if name == "SCAN":
            h &= ~0x7F800  # bits 11-18 of a scan bundle's header are its own (kind, result bits): the checks of the common bits skip them
        assert (h >> 19 == 0 or name == "SCAN") and (1 <= cnt <= G or (cnt == 0 and name in ("LIN", "SYNC")))
        assert (name == "SYNC") == bool(h & (HDR_POST | HDR_WAIT)) and not (name == "SYNC" and cnt)
        a_canon, b_canon, out_canon = bool(h & HDR_A_CANON), bool(h & HDR_B_CANON), bool(h & HDR_OUT_CANON)
        assert not (a_canon or b_canon) or name in ("BIT", "IDIVMOD", "CMPS")
        assert not out_canon or name in ("BIT", "IDIVMOD", "CMPS", "CMPZ", "INPUT")
        one_out = 1 if out_canon else R_MONT
        assert bool(h & HDR_WAIT) == (blob.n_streams > 1 and stream != 0 and b == blob.stream_first[stream])
        assert not (h & HDR_POST) or (stream == 0 and blob.n_streams > 1)
        # narrow multiplication bundle: four lanes per product, a node's record sits at positions 4j .. 4j+3 and its
        # value t + T * j is staged by lane 4 * T * j + t
        rep = COOP_LANES if name in ("MULQ", "MULF") else 1
        assert name != "MULQ" or (T <= COOP_MAX_T and cnt * rep <= G)
        assert name != "MULF" or (T <= COOP_FUSE_MAX_T and cnt * rep <= G)
        stage = LDS_STAGE_OFF + (b % OPND_AHEAD) * STAGE_BYTES
        results = []
        lin_seen = 0
        fused_bits = 0
        if name == "DIVREQ":
            assert blob.divider and mailbox is None, "one division request in flight at a time"
            request = {}
        if name == "DIVGET":
            assert blob.divider and mailbox is not None and len(mailbox) == cnt, "collect must mirror the request"
        def fetch(off, la, q, rec_pos):
            """operand q (0: a fields, 1: b fields) of the record at position rec_pos: its own stage cell or a ring cell"""
            own_cell = stage + 2 * q * LDS_HALF_BYTES + rec_pos * T * 16
            if la == own_cell:  # memory operand, staged OPND_AHEAD bundles ahead
                return mem_at(off, b - OPND_AHEAD - 1, stream, b - OPND_AHEAD)
            assert off == zero_off, "ring operand must stage the zero constant"
            rs, rem = divmod(la - LDS_RING_OFF, RING_SLOT_BYTES)
            assert 0 <= rs < RING_BUNDLES and rem < LDS_HALF_BYTES and rem % (16 * T) == 0
            wb, val = ring[(rs, rem // (16 * T))]
            assert 1 <= b - wb <= RING_BUNDLES, "ring cell too old"
            return val
        if name == "SCAN":
            h = h_scan
            # The steps of serial limb recurrences in consecutive pairs of positions, chain segments one behind the other: a
            # step takes the accumulator of the pair in front of it unless its START bit says "my own operand".  All values
            # are canonical integers; the arithmetic is the unfused nodes' (x + acc, Band / Shr; acc * 2^k + x, Idiv / Mod).
            is_div, sh, iters = bool(h & HDR_SCAN_DIV), (h >> HDR_SCAN_SHIFT_SHIFT) & 0xFF, (h >> HDR_SCAN_ITER_SHIFT) + 1
            is_conv = bool(h & HDR_SCAN_CONV)
            assert T <= SCAN_MAX_T and blob.stats["class_bundles"][CLASS_NAMES.index("MULF")] == 0
            if is_conv:
                # The 2k - 1 columns of a k x k limb product: position c names x_c and y_c (c < k), its result is the sum of
                # x_i y_j with i + j = c, in the field on canonical integers (the unfused Mul / Add nodes' arithmetic).
                k = iters
                assert cnt == 2 * k - 1 and k >= 2 and sh == 0 and not is_div and (h & 0x7E800) == 0
                xs, ys, dsts = [], [], []
                for c in range(cnt):
                    r_ = blob.recs[(b * G + c) * 4:(b * G + c) * 4 + 4]
                    assert (r_[2] & CTRL_MASK) == CTRL_ACTIVE
                    xs.append(fetch(r_[0], r_[3] & 0xFFFF, 0, c))
                    ys.append(fetch(r_[1], r_[3] >> 16, 1, c))
                    dsts.append(r_[2] & ~CTRL_MASK)
                assert all(v < model.M for v in xs + ys)
                for c in range(cnt):
                    results.append((dsts[c], sum(xs[i] * ys[c - i] for i in range(k) if 0 <= c - i < k) % model.M))
            is_sel = (h & (HDR_SCAN_BORROW | HDR_SCAN_LEX)) == (HDR_SCAN_BORROW | HDR_SCAN_LEX)  # selections: the comparison's code in the shift field
            is_borrow, is_lex = bool(h & HDR_SCAN_BORROW) and not is_sel, bool(h & HDR_SCAN_LEX) and not is_sel
            if is_sel:
                assert not is_div and not is_conv and iters == 1 and 1 <= (sh & 7) <= 5 and sh < 16 and not (h & (HDR_SCAN_KG | HDR_SCAN_KL))
                for pr in range(cnt // 2):
                    ro = blob.recs[(b * G + 2 * pr) * 4:(b * G + 2 * pr) * 4 + 4]
                    ra = blob.recs[(b * G + 2 * pr + 1) * 4:(b * G + 2 * pr + 1) * 4 + 4]
                    co, ca = ro[2] & CTRL_MASK, ra[2] & CTRL_MASK
                    assert co & CTRL_ACTIVE and ca & CTRL_ACTIVE and not (co & SCAN_ROLE_ACC) and (ca & SCAN_ROLE_ACC) and (co & SCAN_START) and (ca & SCAN_START)
                    xa = fetch(ro[0], ro[3] & 0xFFFF, 0, 2 * pr)
                    pa, qa = fetch(ra[0], ra[3] & 0xFFFF, 0, 2 * pr + 1), fetch(ra[1], ra[3] >> 16, 1, 2 * pr + 1)
                    code = sh & 7
                    if code == 5:   # the condition's stored word against zero (either form)
                        assert ro[1] == zero_off
                        cond = xa != 0
                    else:           # an ordered comparison of canonical integers (graph.rs:130-133)
                        xb = fetch(ro[1], ro[3] >> 16, 1, 2 * pr)
                        assert xa < model.M and xb < model.M
                        cond = bool(model.eval_duo(["Lt", "Gt", "Leq", "Geq"][code - 1], xa, xb))
                    results.append((ro[2] & ~CTRL_MASK, (R_MONT if sh & 8 else 1) if cond else 0))
                    results.append((ra[2] & ~CTRL_MASK, pa if cond else qa))
            assert is_conv or (cnt % 2 == 0 and (h & 0x61000) == 0 and sh < 254 and is_div + is_borrow + is_lex <= 1)
            assert is_lex or not (h & (HDR_SCAN_KG | HDR_SCAN_KL))
            assert not is_lex or sh <= 1  # (1: the chain's bits are Montgomery-form booleans)
            acc, seg, longest = None, 0, 0
            for pr in range(0 if is_conv or is_sel else cnt // 2):
                ro = blob.recs[(b * G + 2 * pr) * 4:(b * G + 2 * pr) * 4 + 4]
                ra = blob.recs[(b * G + 2 * pr + 1) * 4:(b * G + 2 * pr + 1) * 4 + 4]
                co, ca = ro[2] & CTRL_MASK, ra[2] & CTRL_MASK
                assert co & CTRL_ACTIVE and ca & CTRL_ACTIVE and not (co & SCAN_ROLE_ACC) and (ca & SCAN_ROLE_ACC) and not ((co | ca) & 4) and (co ^ ca) & SCAN_START == 0
                x = fetch(ro[0], ro[3] & 0xFFFF, 0, 2 * pr)
                if co & SCAN_START:
                    acc, seg = fetch(ro[1], ro[3] >> 16, 1, 2 * pr), 1
                else:
                    assert acc is not None and ro[1] == zero_off, "a step without a chain in front of it"
                    seg += 1
                longest = max(longest, seg)
                assert x < model.M and acc < model.M
                if is_div:
                    d, bm = fetch(ra[0], ra[3] & 0xFFFF, 0, 2 * pr + 1), fetch(ra[1], ra[3] >> 16, 1, 2 * pr + 1)
                    assert bm == (1 << sh) * R_MONT % model.M, "the base operand is 2^k in Montgomery form"
                    t = (acc * (1 << sh) + x) % model.M
                    out, acc = (t // d, t % d) if d else (0, 0)
                elif is_borrow or is_lex:
                    y = fetch(ra[0], ra[3] & 0xFFFF, 0, 2 * pr + 1)
                    one_bit = R_MONT if is_lex and sh else 1
                    assert ra[1] == zero_off and y < model.M and (acc in (0, one_bit) or (co & SCAN_START and acc in (0, 1))), "the accumulators of the one-bit recurrences are bits"
                    acc = 1 if acc else 0
                    if is_borrow:  # the unfused nodes' arithmetic: s = y + bin; c = x >= s (signed, graph.rs:133); the two arms in the field
                        c = model.eval_duo("Geq", x, (y + acc) % model.M)
                        out, acc = ((x - y - acc) % model.M if c else (x - y - acc + (1 << sh)) % model.M), (0 if c else 1)
                    else:
                        gt, lt = model.eval_duo("Gt", x, y), model.eval_duo("Lt", x, y)
                        out = None  # (the inner selection: read by nothing)
                        acc = ((1 if h & HDR_SCAN_KG else 0) if gt else (1 if h & HDR_SCAN_KL else 0) if lt else acc) * one_bit
                else:
                    assert ra[0] == zero_off and ra[1] == zero_off
                    t = (x + acc) % model.M
                    out, acc = t & ((1 << sh) - 1), t >> sh
                if out is None:
                    assert (ro[2] & ~CTRL_MASK) == trash, "a comparison step's OUT value has no slot"
                    out = 0
                results.append((ro[2] & ~CTRL_MASK, out))
                results.append((ra[2] & ~CTRL_MASK, acc))
            assert is_conv or is_sel or iters == longest, "the iteration count is the longest chain segment"
            for pos in range(cnt, G):
                a_off, b_off, dctl, lds = blob.recs[(b * G + pos) * 4:(b * G + pos) * 4 + 4]
                assert not (dctl & CTRL_ACTIVE) and (dctl & ~CTRL_MASK) == trash and a_off == zero_off and b_off == zero_off
        for pos in range(0 if name == "SCAN" else G):
            j = pos // rep
            a_off, b_off, dctl, lds = blob.recs[(b * G + pos) * 4:(b * G + pos) * 4 + 4]
            if name == "MULF" and pos % rep:
                # positions 4j+1 / 4j+3: the extra record (operands of the second / third stage, op3), 4j+2: the main record again
                assert blob.recs[(b * G + pos) * 4:(b * G + pos) * 4 + 4] == blob.recs[(b * G + j * rep + (pos & 1)) * 4:(b * G + j * rep + (pos & 1)) * 4 + 4], "a fused node's records alternate main / extra"
                continue
            if pos % rep:
                assert blob.recs[(b * G + pos) * 4:(b * G + pos) * 4 + 4] == blob.recs[(b * G + j * rep) * 4:(b * G + j * rep) * 4 + 4], "the lanes of a product share one record"
                continue
            ctrl, dst = dctl & CTRL_MASK, dctl & ~CTRL_MASK
            assert bool(ctrl & CTRL_ACTIVE) == (j < cnt)
            ops = []
            bitx = name == "BIT" and (ctrl & CTRL_SUB_MASK) == 5 and j < cnt  # (a >> k) & 1 with k = b_lds / 16
            for q, (off, la) in enumerate(((a_off, lds & 0xFFFF), (b_off, lds >> 16))):
                if bitx and q == 1:
                    assert off == zero_off and la % 16 == 0 and la // 16 < 254
                    ops.append(la // 16)
                    continue
                ops.append(fetch(off, la, q, j * rep))
            if j >= cnt:  # padding: harmless operands, store to the trash slot
                assert dst == trash and a_off == zero_off and b_off == zero_off
                if name == "MULF":
                    assert (ctrl & CTRL_SUB_MASK) == 0 and (blob.recs[(b * G + pos + 1) * 4 + 2] & CTRL_SUB_MASK) == 0, "idle groups of a fused bundle have no later stage"
                continue
            sub = ctrl & CTRL_SUB_MASK
            if name == "MULF":
                # (a * b) op2 x2 op3 x3 on the stored words: the same Montgomery products / modular sums as the unfused nodes
                xa_off, xb_off, xdctl, xlds = blob.recs[(b * G + pos + 1) * 4:(b * G + pos + 1) * 4 + 4]
                op2, op3 = sub, xdctl & CTRL_SUB_MASK
                assert op2 <= FOP_RSUB and op3 in (FOP_NONE, FOP_ADD, FOP_SUB, FOP_RSUB) and (xdctl & ~CTRL_MASK) == trash and (xdctl & CTRL_ACTIVE)
                fused_bits |= (HDR_F_S2MUL if op2 == FOP_MUL else HDR_F_S2LIN if op2 else 0) | (HDR_F_S3LIN if op3 else 0)
                acc = ops[0] * ops[1] * R_INV % model.M

                def stage_op(code, acc, x):
                    if code == FOP_MUL:
                        return acc * x * R_INV % model.M
                    if code == FOP_ADD:
                        return (acc + x) % model.M
                    return (acc - x) % model.M if code == FOP_SUB else (x - acc) % model.M
                if op2:
                    acc = stage_op(op2, acc, fetch(xa_off, xlds & 0xFFFF, 0, j * rep + 1))
                else:
                    assert xa_off == zero_off
                if op3:
                    acc = stage_op(op3, acc, fetch(xb_off, xlds >> 16, 1, j * rep + 1))
                else:
                    assert xb_off == zero_off
                results.append((dst, acc))
                continue

            def mont_div(x, y):  # fr_inv: Montgomery in, Montgomery out; b == 0 -> 0 (graph.rs:109)
                return x * pow(y, -1, model.M) * R_MONT % model.M if y % model.M else 0
            if name == "DIVREQ":
                request[j] = (ops[0], ops[1])
                assert dst == trash
                v = 0
            elif name == "DIVGET":
                v = mont_div(*mailbox[j])
            elif name == "INPUT":
                v = inputs_row[blob.crefs[cref_row * G + j]] % model.M * (1 if out_canon else R_MONT) % model.M  # Fr::new, kept canonical in bit graphs
            elif name == "TERN":
                v = mem_at(blob.crefs[cref_row * G + j], b - 1, stream, b) if ops[0] == 0 else ops[1]
            else:
                op = SUB_NAMES[name][sub]
                assert op is not None
                if name in ("LIN", "MUL", "MULQ"):
                    if op != "Mul":
                        lin_seen |= (1 << 11) if op == "Sub" else (1 << 12)
                        v = model.eval_duo(op, ops[0], ops[1])  # (a +- b mod r: the same words in either form)
                    elif name == "MUL" and h & HDR_MUL_CC:
                        v = ops[0] * ops[1] % model.M            # canonical x canonical -> canonical
                    else:
                        v = ops[0] * ops[1] * R_INV % model.M    # Montgomery product
                elif name == "DIV":
                    v = mont_div(ops[0], ops[1])
                elif name == "CMPZ":  # zero tests and equality of the stored words; both sides are in one form
                    v = one_out if model.eval_duo(op, ops[0], ops[1]) else 0
                else:  # BIT / IDIVMOD / CMPS: on the canonical integers
                    x = ops[0] if a_canon else ops[0] * R_INV % model.M
                    if op == "BitX":
                        results.append((dst, one_out if (x >> ops[1]) & 1 else 0))
                        continue
                    y = ops[1] if b_canon else ops[1] * R_INV % model.M
                    try:
                        d = model.eval_duo(op, x, y)
                    except model.ReferencePanic:
                        status |= 1 if op == "Shl" else 2
                        d = 0
                    v = d if out_canon else d * R_MONT % model.M
            results.append((dst, v))
        cref_row += name in ("INPUT", "TERN")
        if name == "DIVREQ":
            mailbox = request
            assert blob.div_lanes[n_requests_total] == cnt * T <= (64 if blob.divider == 1 else 32), "request must fit the mailbox"
            n_requests += 1
            n_requests_total += 1
        elif name == "DIVGET":
            mailbox = None
        else:
            assert name != "DIV" or not blob.divider
        if name in ("MULF", "SCAN"):
            pass
        elif name == "BIT":
            all_x = all((blob.recs[(b * G + jj) * 4 + 2] & CTRL_SUB_MASK) == 5 for jj in range(cnt))
            assert ((h >> 13) & 1) == (1 if all_x else 0), "BITX header bit must describe the records"
            subs = [blob.recs[(b * G + jj) * 4 + 2] & CTRL_SUB_MASK for jj in range(cnt)]
            limb = all(s_ in (1, 3) for s_ in subs)  # Shr and Band nodes only: bit 11 if any shifts, else bit 12
            lin_seen = 0 if not limb else (1 << 11) if 1 in subs else (1 << 12)
        elif name == "MUL":
            assert not (h & HDR_MUL_CC) or (lin_seen == 0 and T <= SCAN_MAX_T), "canonical products: no riders, tile widths with the MODE 2 instances"
        else:
            assert ((h >> 13) & 1) == 0
        if name == "MULF":
            assert (h & (HDR_F_S2MUL | HDR_F_S2LIN | HDR_F_S3LIN)) == fused_bits, "stage bits of a fused bundle must describe its records"
        elif name != "SCAN":
            assert ((h >> 11) & 3) == (lin_seen >> 11), "LIN header bits must describe the records"
        dsts = [d for d, _ in results if d != trash]
        assert len(set(dsts)) == len(dsts), "two nodes of one bundle share a destination slot"
        for d, v in results:
            assert d == trash or (d % slot_bytes == 0 and NC * slot_bytes <= d < (NC + blob.n_slots) * slot_bytes)
            if d != trash:
                hist = history.setdefault(d // slot_bytes - NC, [])
                assert not hist or hist[0][0] == stream, "a slot is written by one stream only"
                hist.append((stream, b, v))
        for j, (_, v) in enumerate(results):
            ring[(b % RING_BUNDLES, j)] = (b, v)
        for j in range(cnt if name != "DIVREQ" else 0, G):
            ring.pop((b % RING_BUNDLES, j), None)  # inactive lanes overwrite the cell with garbage

        if b == blob.stream_first[stream] + blob.stream_count[stream] - 1:
            assert mailbox is None and n_requests == blob.stream_div_requests[stream]
    assert n_requests_total == blob.n_div_requests
    for slot in cross_reads:
        assert history[slot][-1][1] <= post_at - 1, "a slot that other streams read is written again behind the post"

    def wit(r):
        if r & REF_CONST:
            return blob.consts[r & 0x3FFFFFFF]
        raw = history[r & ~REF_CANON][-1][2]
        return raw if r & REF_CANON else raw * R_INV % model.M
    return [wit(r) for r in blob.witness_refs], status
