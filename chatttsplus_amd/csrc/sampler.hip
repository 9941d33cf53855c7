// Fused sampler: one wavefront per (sequence, codebook) row of 626 logits (10 per lane).
//
// Restates, in order (chattts_plus/models/gpt.py:469-494,527-532):
//   logits /= temperature                                              gpt.py:469
//   CustomRepetitionPenaltyLogitsProcessorRepeat (window 16)            models/processors.py:18-34
//   TopPLogitsWarper(top_p, min_tokens_to_keep=3)   (transformers; built at processors.py:45)
//   TopKLogitsWarper(top_k, min_tokens_to_keep=3)   (transformers; built at processors.py:47)
//   i < min_new_token -> logits[eos] = -inf                              gpt.py:477-478
//   softmax ; multinomial(1) == argmax(p / q), q ~ Exp(1)               gpt.py:480-481
//   finish |= any(idx == eos); ids_buf[progress] = idx; end_idx += ~finish     gpt.py:483-488,530-531
//   next-token embedding = sum of the 4 code embeddings                  gpt.py:403-407
//
// Top-p needs the *ascending* cumulative softmax.  Because the removed set is a prefix of the ascending
// order, and top-k follows, only the largest <= top_k (+ties) candidates are ever kept: they are found with a
// value threshold + rank-by-counting (sample_row below), and the ascending cumsum at rank r is
// total - (sum of the r larger probabilities), accumulated in fp64 exactly like torch's CPU cumsum
// (acc_type<float> = double) and rounded to fp32 before the `<= 1 - top_p` compare.
#include "kernels.h"

#define VPL 10   // values per lane (64 * 10 = 640 >= 626)

__device__ inline uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}

struct RowIn {
    const float* q;          // [V] or null -> Philox
    int myid;                // lane hh < 16: id hh of the repetition-penalty window, or -1 (not in the window)
    float T;
    bool penalize;
    int step;
    unsigned long long seed;
    unsigned uid_lo, uid_hi;   // Philox stream of the row: (seed | utterance id, codebook, the row's own step and regenerate attempt)
    unsigned vq, attempt;
};

struct SampleKnobs { float top_p_threshold; int top_k, min_keep, eos, min_new; };
__device__ inline SampleKnobs knobs_of(SamplerDynPtr d) {
    SampleKnobs k;
    k.top_p_threshold = d->cfg.top_p_threshold; k.top_k = d->cfg.top_k; k.min_keep = d->cfg.min_keep; k.eos = d->cfg.eos; k.min_new = d->cfg.min_new;
    return k;
}

// Exp(1) noise of element j of one multinomial row, counter-based: Philox4x32-10 keyed by the request seed, counter = (element | stream << 24,
// utterance id, step | attempt << 20), stream = codebook 0..3 or 4 for the refine-text row.  Nothing in it depends on the row's place in a
// batch, on the batch's size or on the batch's draw counter: an utterance draws the same noise in whatever slice / rank / decode row it is
// served.  u = (24 random bits + 0.5) / 2^24 in (0, 1); restated in oracle/device_noise.py and pinned through ctts_sampler_noise.
__device__ inline float device_exp_noise(unsigned long long seed, unsigned uid_lo, unsigned uid_hi, unsigned stream, unsigned step, unsigned attempt, int j) {
    const uint4 rnd = philox4x32_10(make_uint4((unsigned)j | (stream << 24), uid_lo, uid_hi, step | (attempt << 20)), make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
    const float u = ((float)(rnd.x >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return -logf(u);
}
__device__ inline float exp_noise_of(const RowIn& in, int j) {
    if (in.q != nullptr) return in.q[j];
    return device_exp_noise(in.seed, in.uid_lo, in.uid_hi, in.vq, (unsigned)in.step, in.attempt, j);
}

// One row = one wavefront.  Selection of the survivors of top-p / top-k:
//   Top-p removes a prefix of the ASCENDING order and top-k follows, so the survivors are a prefix of the DESCENDING order, at
//   most top_k (+ ties with the k-th value) long.  Instead of pulling them out one by one (the former <= 20+ dependent wave-wide
//   arg-max rounds, ~8 us of a 16 us kernel), a value threshold T0 <= (k-th largest value) is found from the 64 lane maxima
//   (T0 = their k-th largest: at least k values are >= T0), the candidates {x >= T0} (typically 20-30) are compacted into one
//   per lane, ranked by counting and permuted into descending order; the ascending cumsum at rank r = total - (sum of the
//   larger probabilities), accumulated in fp64 in rank order exactly as before, decides the cut for all ranks at once.
//   Falls back to the serial extraction when top_k > 64 or more than 64 candidates tie at the threshold.
// `lds64`: 64 x 8 bytes of LDS private to this wave.
__device__ int sample_row(const SampleKnobs& c, const float* tab, const RowIn& in, int V, int lane, const float (&lg)[VPL], unsigned long long* lds64) {
    float x[VPL];
    unsigned valid = 0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int j = lane + 64 * i;
        if (j < V) { x[i] = __fdiv_rn(lg[i], in.T); valid |= 1u << i; }
        else x[i] = -INFINITY;
    }
    if (in.penalize) {
        // window ids live one per lane (lane hh < 16); each is broadcast and counted into a packed 5-bit counter per slot
        unsigned long long cntp = 0ull;
#pragma unroll
        for (int hh = 0; hh < 16; ++hh) {
            const int id = __builtin_amdgcn_readlane(in.myid, hh);
            const unsigned long long inc = 1ull << (5 * ((unsigned)id >> 6) & 63);      // id == -1 -> owner lane 63, slot >= 10: never read
            cntp += (id >= 0 && (id & 63) == lane) ? inc : 0ull;
        }
        if (__builtin_amdgcn_ballot_w64(cntp != 0ull) != 0ull) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const float alpha = tab[(unsigned)(cntp >> (5 * i)) & 31u];
                x[i] = (x[i] < 0.f) ? __fmul_rn(x[i], alpha) : __fdiv_rn(x[i], alpha);   // processors.py:29-33 (alpha == 1 leaves x unchanged)
            }
        }
    }
    // softmax over the whole row (TopPLogitsWarper: sorted_logits.softmax(-1))
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPL; ++i) mx = fmaxf(mx, x[i]);
    mx = wave_max(mx);
    float pr[VPL], se = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { pr[i] = (valid >> i & 1u) ? expf(x[i] - mx) : 0.f; se += pr[i]; }
    se = wave_sum(se);
    const float inv = 1.0f / se;
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { pr[i] *= inv; tot += (double)pr[i]; }
    tot = wave_sum_d(tot);

    const int topk = (c.top_k > 0) ? c.top_k : V;
    float myv = -INFINITY;                                         // lane r: the r-th selected element (descending order)
    int myi = 0, nsel = 0;
    unsigned kept = 0;                                             // serial path only
    bool fast = false;
    if (topk <= 64) {
        unsigned vkey[VPL], head = 0u;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { vkey[i] = ((valid >> i) & 1u) ? f32_key(x[i]) : 0u; head = max(head, vkey[i]); }
        int greater = 0;                                           // lane maxima strictly above mine
#pragma unroll
        for (int j = 0; j < 64; ++j) greater += ((unsigned)__builtin_amdgcn_readlane((int)head, j) > head) ? 1 : 0;
        // k-th largest lane maximum = the smallest one with fewer than k maxima above it
        const unsigned long long t0k = wave_max_u64((greater < topk) ? (unsigned long long)(~head) : 0ull);
        const unsigned T0 = ~(unsigned)t0k;
        int C = 0;
        int pos[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const bool cand = ((valid >> i) & 1u) && vkey[i] >= T0;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(cand);
            pos[i] = cand ? C + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u)) : -1;
            C += __builtin_popcountll(bal);
        }
        if (C <= 64) {
            fast = true;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (pos[i] >= 0) lds64[pos[i]] = ((unsigned long long)vkey[i] << 32) | (unsigned)(lane + 64 * i);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned long long ck = (lane < C) ? lds64[lane] : 0ull;
            int rank = 0;                                          // keys are distinct (index in the low word): ranks = a permutation of 0..C-1
            for (int j = 0; j < C; ++j) rank += (readlane_u64(ck, j) > ck) ? 1 : 0;
            // lane r <- the candidate of rank r (lanes >= C push zeros to lane C: harmless)
            const unsigned lo = (unsigned)__builtin_amdgcn_ds_permute(rank << 2, (int)(unsigned)ck);
            const unsigned hi = (unsigned)__builtin_amdgcn_ds_permute(rank << 2, (int)(unsigned)(ck >> 32));
            const float bv = key_f32(hi);
            const int bi = (int)lo;
            const float p_r = (lane < C) ? expf(bv - mx) * inv : 0.f;     // the same fp32 expression that produced pr[] for this element
            const float vk = readlane_f(bv, (topk - 1) & 63);             // only consulted at ranks >= topk, which exist only if C > topk - 1
            double cum = 0.0, cum_mine = 0.0;                             // cum_mine = sum of the larger probabilities, rank order, fp64
            for (int r = 0; r < C; ++r) {
                if (lane == r) cum_mine = cum;
                cum += (double)readlane_f(p_r, r);
            }
            const float cr = (float)(tot - cum_mine);                     // ascending cumsum at this element
            const bool stop = (lane >= C) || (lane >= topk && bv != vk) || (cr <= c.top_p_threshold && lane >= c.min_keep);
            const unsigned long long sb = __builtin_amdgcn_ballot_w64(stop);
            nsel = (sb == 0ull) ? 64 : (int)__builtin_ctzll(sb);
            myv = bv; myi = bi;
        }
    }
    if (!fast) {
        // serial extraction: each lane sorts its 10 (value,index) keys once (descending, odd-even transposition network); every
        // round compares the 64 lane heads (one DPP arg-max) and the winner's lane pops its head
        unsigned long long key[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            key[i] = ((valid >> i) & 1u) ? (((unsigned long long)f32_key(x[i]) << 32) | (unsigned)(lane + 64 * i)) : 0ull;
#pragma unroll
        for (int pass = 0; pass < VPL; ++pass)
#pragma unroll
            for (int i = pass & 1; i + 1 < VPL; i += 2) {
                const unsigned long long a = key[i], b = key[i + 1];
                key[i] = a > b ? a : b;
                key[i + 1] = a > b ? b : a;
            }
        double cum_before = 0.0;
        float vk = 0.f;
        for (int r = 0; r < V; ++r) {
            const unsigned long long best = wave_max_u64(key[0]);     // larger index wins ties (== reversed stable ascending sort)
            if (best == 0ull) break;                                   // nothing left
            const float bv = key_f32((unsigned)(best >> 32));
            const int bi = (int)(unsigned)best;
            const int owner = bi & 63, slot = bi >> 6;
            if (r >= topk && bv != vk) break;                          // beyond top-k and not tied with the k-th value
            const float cr = (float)(tot - cum_before);               // ascending cumsum at this element
            if (cr <= c.top_p_threshold && r >= c.min_keep) break;     // removed by top-p (and so is every smaller one)
            if (lane == owner) {
                kept |= 1u << slot;
#pragma unroll
                for (int i = 0; i + 1 < VPL; ++i) key[i] = key[i + 1];
                key[VPL - 1] = 0ull;
            }
            if (lane == r) { myv = bv; myi = bi; }
            nsel = r + 1;
            if (r == topk - 1) vk = bv;
            cum_before += (double)(expf(bv - mx) * inv);               // the same fp32 expression that produced pr[] for this element
        }
        if (in.step < c.min_new) {                                     // gpt.py:477-478
            if (lane == (c.eos & 63)) kept &= ~(1u << (c.eos >> 6));
        }
    }
    // final softmax over the kept set and the exponential race
    unsigned long long bkey = 0ull;                                // (ratio, first index wins ties) as one 64-bit key
    if (nsel <= 64) {
        // usual case (top-k / top-p keep a handful): one kept element per lane -- one exp, one noise draw, one division per
        // lane.  Elements outside the kept set have p == 0 -> ratio 0: element 0 stands in for all of them (smallest index
        // wins ties).
        const bool on = (lane < nsel) && !(in.step < c.min_new && myi == c.eos);
        const float m2 = wave_max(on ? myv : -INFINITY);
        const float e = on ? expf(myv - m2) : 0.f;
        const float inv2 = 1.0f / wave_sum(e);
        bkey = ((unsigned long long)f32_key(0.f) << 32) | (unsigned)0x7FFFFFFF;
        if (on) {
            const float ratio = __fdiv_rn(e * inv2, exp_noise_of(in, myi));
            bkey = umax64(bkey, ((unsigned long long)f32_key(ratio) << 32) | (unsigned)(0x7FFFFFFF - myi));
        }
    } else {
        float m2 = -INFINITY;
#pragma unroll
        for (int i = 0; i < VPL; ++i) if ((kept >> i) & 1u) m2 = fmaxf(m2, x[i]);
        m2 = wave_max(m2);
        float e2[VPL], s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { e2[i] = ((kept >> i) & 1u) ? expf(x[i] - m2) : 0.f; s2 += e2[i]; }
        s2 = wave_sum(s2);
        const float inv2 = 1.0f / s2;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int j = lane + 64 * i;
            if (j < V) {
                // elements outside the kept set have p == 0 -> ratio 0 whatever q is: only kept ones draw noise
                float ratio = 0.f;
                if ((kept >> i) & 1u) ratio = __fdiv_rn(e2[i] * inv2, exp_noise_of(in, j));
                bkey = umax64(bkey, ((unsigned long long)f32_key(ratio) << 32) | (unsigned)(0x7FFFFFFF - j));
            }
        }
    }
    bkey = wave_max_u64(bkey);
    const int besti = 0x7FFFFFFF - (int)(unsigned)bkey;
    return besti;
}

// generate mode: grid = B blocks, block = 4 waves (one per codebook)
// The leading scalars are preloaded into SGPRs (see skinny_gemm.hip): state header, logits and bookkeeping rows are all
// requested before the first wait of the kernel.
__global__ __launch_bounds__(256) void sampler_generate_kernel(const int* st_words, const float* logits_p, const SamplerDyn* dyn_p, const RowMeta* meta_p,
                                                               const int V_p, const int* ring_p, const RowState* finend_p, const SamplerArgs a) {
    __shared__ float tab[17];
    __shared__ int idx_s[CTTS_NUM_VQ];
    __shared__ unsigned long long cand_s[CTTS_NUM_VQ][64];
    DevState* st = a.st;
    const SamplerDynPtr d = (SamplerDynPtr)dyn_p;
    const int tid = threadIdx.x, lane = tid & 63, vq = tid >> 6;
    const int b = blockIdx.x;
    int zoff;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
    const int4 hdr = *(const int4*)(st_words + zoff);             // DevState {step, draw, all_done, ticket}: one vector load
    float lg[VPL];
    {
        const float* lrow = logits_p + (size_t)(b * CTTS_NUM_VQ + vq) * V_p;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { const int j = lane + 64 * i; lg[i] = (j < V_p) ? lrow[j] : 0.f; }
    }
    // Everything the kernel reads lives at addresses known at launch (preloaded kernel arguments), so ALL its loads leave in one
    // batch: the repetition-penalty window is a 16-entry ring per (sequence, codebook) that this kernel maintains itself (the ids
    // buffer's address depends on the step and on the per-call block: a second dependent round trip in front of the penalty), and
    // the finish / end_idx bookkeeping is mirrored in engine memory (the caller's arrays are write-only here).
    const int ring_v = ring_p[(size_t)(b * CTTS_NUM_VQ + vq) * 16 + (lane & 15)];
    const int4 fe = ((const int4*)(finend_p + b))[0];              // {fin, end, attempt, limit}
    const int4 uid = ((const int4*)(finend_p + b))[1];             // {uid_lo, uid_hi, out, -}
    if (tid < 17) tab[tid] = d->cfg.penalty_table[tid];
    const RowMeta meta_in = meta_p[b];
    if (__builtin_amdgcn_readfirstlane(hdr.z)) return;            // every sequence finished (gpt.py:545)
    const int gstep = __builtin_amdgcn_readfirstlane(hdr.x), draw = __builtin_amdgcn_readfirstlane(hdr.y);
    const int fin_in = fe.x, end_in = fe.y;
    // the row's OWN step (i of gpt.py:389 for this utterance) = tokens it has sampled so far: equal to the batch's step counter while the
    // row is live and was part of the batch from its start; rows admitted later (ctts_gpt_admit) run behind it
    const int step = end_in;
    const int seq = uid.z;                                        // utterance of this row: output arrays, noise rows (rows are re-packed by ctts_gpt_compact)
    const float rope_next = (tid < 64) ? a.rope[(size_t)(meta_in.pos + 1) * 64 + tid] : 0.f;
    __syncthreads();
    const int row = seq * CTTS_NUM_VQ + vq;                       // row of the [B0 * 4, V] batch the call started with (gpt.py:444-447)
    RowIn in;
    in.q = (d->noise != nullptr) ? d->noise + ((size_t)min(draw, d->n_draws - 1) * d->rows0 + row) * a.V : nullptr;
    // ring slot p holds the id sampled at the latest step s < `step` with s % 16 == p: it is inside the window of the last
    // min(step, past_window) ids (processors.py:21-23 with gpt.py:455-457) iff its age step - s is at most that
    const int nh = min(step, d->cfg.past_window);
    const int age = ((step - 1 - lane) & 15) + 1;
    in.myid = (lane < 16 && age <= nh) ? ring_v : -1;
    in.T = d->cfg.temperature[vq];
    // quirk SURVEY F8: the reference zeroes the penalty for rows >= max_input_ids of the flattened [B * 4] batch IT runs -- the row's place in the
    // decode batch (b * 4 + vq <= 511), not the utterance's place in the caller's output arrays (`row`, which grows without bound under
    // ctts_gpt_admit / generate_many and silently switched the penalty off from utterance 157 on)
    in.penalize = d->cfg.use_penalty && (b * CTTS_NUM_VQ + vq < d->cfg.max_input_ids) && nh > 0;
    in.step = step;
    in.seed = d->seed; in.uid_lo = (unsigned)uid.x; in.uid_hi = (unsigned)uid.y; in.vq = (unsigned)vq; in.attempt = (unsigned)fe.z;
    const int idx = sample_row(knobs_of(d), tab, in, a.V, lane, lg, cand_s[vq]);
    if (lane == 0) {
        idx_s[vq] = idx;
        if (fin_in == 0) d->ids[((size_t)seq * d->cfg.max_new + step) * CTTS_NUM_VQ + vq] = idx;      // (a finished row's tokens are never read: gpt.py:295-297)
        a.hist_ring[(size_t)(b * CTTS_NUM_VQ + vq) * 16 + (step & 15)] = idx;
    }
    __syncthreads();
    // next-token embedding: ((e0 + e1) + e2) + e3   (torch.stack(..., 3).sum(3), gpt.py:403-407)
    for (int k = tid; k < a.H; k += 256) {
        float s = a.emb_code[((size_t)0 * a.V + idx_s[0]) * a.H + k];
#pragma unroll
        for (int v = 1; v < CTTS_NUM_VQ; ++v) s += a.emb_code[((size_t)v * a.V + idx_s[v]) * a.H + k];
        a.x_next[(size_t)b * a.H + k] = s;
    }
    if (tid == 0) {
        const bool was = fin_in != 0;
        bool eos = (fin_in & 2) != 0;
        for (int v = 0; v < CTTS_NUM_VQ; ++v) eos = eos || (idx_s[v] == d->cfg.eos);   // gpt.py:486-487
        bool fin = was || eos;
        const int end_out = fin ? end_in : end_in + 1;                               // gpt.py:530-531
        if (!fin) d->end_idx[seq] = end_out;
        fin = fin || (end_out >= fe.w);                                              // the row's own token limit (<= max_new_token: the loop bound of gpt.py:389): done from the next step on
        if (!was) d->finish[seq] = eos ? 1 : 0;                                      // the caller's `finish` keeps the reference's meaning: EOS seen (a row that was already finished
                                                                                     // stores nothing: its `out` may have been handed to a re-admitted attempt of the same utterance)
        ((int2*)(a.finend + b))[0] = make_int2(fin ? (eos ? 3 : 1) : 0, end_out);
        if (!was) {                                                                // next decode row; a finished row stays on its last slot (it keeps
            RowMeta m = meta_in;                                                   // computing like the reference's finished rows, gpt.py:527-546, but
            m.pos += 1; m.slot += 1;                                               // never walks past the end of its cache lane)
            a.meta[b] = m;
        }
        // One agent-scope atomic carries both the arrival ticket (low 16 bits) and the number of finished
        // sequences (high 16 bits, persistent over the steps): no fences, no cross-block plain loads.
        const int add = 1 + ((fin && !was) ? 0x10000 : 0);
        // single sequence: no other block to wait for, no atomic round trip
        const int tot = (a.B == 1) ? (__builtin_amdgcn_readfirstlane(hdr.w) + add) : (__hip_atomic_fetch_add(&st->ticket, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add);
        if ((tot & 0xFFFF) == a.B) {                           // last block of this step: advance the step state
            const int nfin = tot >> 16;
            st->ticket = nfin << 16;
            st->step = gstep + 1;
            st->draw = draw + 1;
            if (nfin == a.B) st->all_done = 1;                 // gpt.py:545 (every row ends by its limit <= max_new_token, the loop bound of :389)
        }
    }
    if (tid < 64) a.rope_rows[(size_t)b * 64 + tid] = rope_next;      // RoPE row of the next step's position (prefetched)
}

// ---- refine-text pass (infer_text=True): one 21178-way row per sequence ------------------------------------------------
// gpt.py:400-401,425-426,458-467,489-494 (called from pipeline:237-277): single temperature, no repetition penalty
// (repetition_penalty == 1 is the only value the reference's processor handles for this mode), top-p then top-k,
// min-length, softmax + multinomial, finish on EOS = [Ebreak], next input = emb_text[id].
// One 1024-thread block per sequence, 21 values per thread; the same "extract the <= top_k largest, fp64 ascending
// cumsum" scheme as the code sampler with block-wide (DPP + LDS) reductions.
#define TVPT 21
struct BlockRed {
    float f[16];
    double d[16];
    unsigned long long k[16];
    unsigned long long k2[2][16];      // alternating slots: one barrier per selection round (block_max_u64_alt)
};
__device__ inline float block_max_f(BlockRed& r, float v, int lane, int wave) {
    v = wave_max(v);
    if (lane == 0) r.f[wave] = v;
    __syncthreads();
    float m = r.f[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, r.f[i]);
    __syncthreads();
    return m;
}
__device__ inline float block_sum_f(BlockRed& r, float v, int lane, int wave) {
    v = wave_sum(v);
    if (lane == 0) r.f[wave] = v;
    __syncthreads();
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) m += r.f[i];
    __syncthreads();
    return m;
}
__device__ inline double block_sum_d(BlockRed& r, double v, int lane, int wave) {
    v = wave_sum_d(v);
    if (lane == 0) r.d[wave] = v;
    __syncthreads();
    double m = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) m += r.d[i];
    __syncthreads();
    return m;
}
__device__ inline unsigned long long block_max_u64(BlockRed& r, unsigned long long v, int lane, int wave) {
    v = wave_max_u64(v);
    if (lane == 0) r.k[wave] = v;
    __syncthreads();
    unsigned long long m = r.k[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = umax64(m, r.k[i]);
    __syncthreads();
    return m;
}

// block-wide max with alternating LDS slots: the write of round r+1 goes to the other slot, so the only barrier needed is the
// one between this round's writes and reads
__device__ inline unsigned long long block_max_u64_alt(BlockRed& r, unsigned long long v, int lane, int wave, int round) {
    v = wave_max_u64(v);
    unsigned long long* slot = r.k2[round & 1];
    if (lane == 0) slot[wave] = v;
    __syncthreads();
    unsigned long long m = slot[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = umax64(m, slot[i]);
    return m;
}

__global__ __launch_bounds__(1024) void sampler_text_kernel(const SamplerArgs a) {
    __shared__ BlockRed red;
    __shared__ int pos_s;
    DevState* st = a.st;
    const SamplerDynPtr d = (SamplerDynPtr)a.dyn;
    if (st->all_done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, V = a.V;
    const int gstep = st->step, draw = st->draw;
    const RowState rs_in = a.finend[b];
    // per-row bookkeeping like the code sampler's (RowState, common.h): the row's OWN step, its utterance's place in the output arrays, its own limit --
    // so text rows can be compacted away and re-used by queued utterances (ctts_gpt_admit) exactly like code rows
    const int step = rs_in.end, seq = rs_in.out;
    const float* lg = a.logits + (size_t)b * V;
    const float* q = (d->noise != nullptr) ? d->noise + ((size_t)min(draw, d->n_draws - 1) * d->rows0 + seq) * V : nullptr;
    const float T = d->cfg.temperature[0];
    float x[TVPT];
    unsigned valid = 0;
#pragma unroll
    for (int i = 0; i < TVPT; ++i) {
        const int j = tid + 1024 * i;
        if (j < V) { x[i] = __fdiv_rn(lg[j], T); valid |= 1u << i; }          // gpt.py:469
        else x[i] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < TVPT; ++i) mx = fmaxf(mx, x[i]);
    mx = block_max_f(red, mx, lane, wave);
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < TVPT; ++i) se += ((valid >> i) & 1u) ? expf(x[i] - mx) : 0.f;
    se = block_sum_f(red, se, lane, wave);
    const float inv = 1.0f / se;
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < TVPT; ++i) tot += ((valid >> i) & 1u) ? (double)(expf(x[i] - mx) * inv) : 0.0;
    tot = block_sum_d(red, tot, lane, wave);

    unsigned taken = ~valid, kept = 0;
    const int topk = (d->cfg.top_k > 0) ? d->cfg.top_k : V;
    // Selection of the top-p / top-k survivors (a prefix of the descending order, <= top_k + ties long).  Fast path, the block-wide
    // version of sample_row's: a value threshold T0 <= (k-th largest value) from the two largest thread maxima of every wave (32
    // distinct elements: their k-th largest bounds the k-th largest overall from below, k <= 32), the candidates {x >= T0} compacted
    // into LDS, ranked and cut by wave 0, and the kept indices handed back to their owner threads -- 4 barriers instead of one
    // block-wide arg-max round (wave reduction + LDS + barrier) per survivor.  Falls back to the serial rounds below.
    __shared__ unsigned top2_s[16][2];
    __shared__ int wtot_s[16];
    __shared__ unsigned long long cand_s[64];
    __shared__ int sel_s[64];
    __shared__ int nsel_s;
    bool fast = false;
    if (topk <= 32) {
        unsigned vkey[TVPT], head = 0u;
#pragma unroll
        for (int i = 0; i < TVPT; ++i) { vkey[i] = ((valid >> i) & 1u) ? f32_key(x[i]) : 0u; head = max(head, vkey[i]); }
        const unsigned long long hk = ((unsigned long long)head << 32) | (unsigned)lane;
        const unsigned long long b1 = wave_max_u64(hk);
        const unsigned long long b2 = wave_max_u64(hk == b1 ? 0ull : hk);
        if (lane == 0) { top2_s[wave][0] = (unsigned)(b1 >> 32); top2_s[wave][1] = (unsigned)(b2 >> 32); }
        __syncthreads();
        const unsigned mine32 = (lane < 32) ? top2_s[lane >> 1][lane & 1] : 0u;
        int greater = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) greater += (top2_s[j >> 1][j & 1] > mine32) ? 1 : 0;
        const unsigned T0 = ~(unsigned)wave_max_u64((lane < 32 && greater < topk) ? (unsigned long long)(~mine32) : 0ull);
        unsigned cmask = 0;
#pragma unroll
        for (int i = 0; i < TVPT; ++i) cmask |= (((valid >> i) & 1u) && vkey[i] >= T0) ? (1u << i) : 0u;
        const int cnt = __builtin_popcount(cmask);
        int incl = cnt;                                              // inclusive prefix over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) wtot_s[wave] = incl;
        __syncthreads();
        int base = 0, C = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { base += (w < wave) ? wtot_s[w] : 0; C += wtot_s[w]; }
        if (C <= 64) {                                               // uniform over the block
            fast = true;
            int pos = base + incl - cnt;
#pragma unroll
            for (int i = 0; i < TVPT; ++i)
                if ((cmask >> i) & 1u) cand_s[pos++] = ((unsigned long long)vkey[i] << 32) | (unsigned)(tid + 1024 * i);
            __syncthreads();
            if (wave == 0) {
                const unsigned long long ck = (lane < C) ? cand_s[lane] : 0ull;
                int rank = 0;
                for (int j = 0; j < C; ++j) rank += (readlane_u64(ck, j) > ck) ? 1 : 0;
                const unsigned lo = (unsigned)__builtin_amdgcn_ds_permute(rank << 2, (int)(unsigned)ck);
                const unsigned hi = (unsigned)__builtin_amdgcn_ds_permute(rank << 2, (int)(unsigned)(ck >> 32));
                const float bv = key_f32(hi);
                const float p_r = (lane < C) ? expf(bv - mx) * inv : 0.f;
                const float vk = readlane_f(bv, (topk - 1) & 63);
                double cum = 0.0, cum_mine = 0.0;
                for (int r = 0; r < C; ++r) {
                    if (lane == r) cum_mine = cum;
                    cum += (double)readlane_f(p_r, r);
                }
                const float cr = (float)(tot - cum_mine);
                const bool stop = (lane >= C) || (lane >= topk && bv != vk) || (cr <= d->cfg.top_p_threshold && lane >= d->cfg.min_keep);
                const unsigned long long sb = __builtin_amdgcn_ballot_w64(stop);
                const int nsel = (sb == 0ull) ? 64 : (int)__builtin_ctzll(sb);
                if (lane < nsel) sel_s[lane] = (int)lo;
                if (lane == 0) nsel_s = nsel;
            }
            __syncthreads();
            const int nsel = nsel_s;
            for (int r = 0; r < nsel; ++r) {
                const int bi = sel_s[r];
                if ((bi & 1023) == tid) kept |= 1u << (bi >> 10);
            }
        }
    }
    if (!fast) {
    double cum_before = 0.0;
    float vk = 0.f;
    // every thread caches the key of its largest untaken value; only the thread whose element was selected rescans its 21 values
    auto my_best = [&]() -> unsigned long long {
        unsigned long long key = 0ull;
#pragma unroll
        for (int i = 0; i < TVPT; ++i) {
            const unsigned long long k = ((unsigned long long)f32_key(x[i]) << 32) | (unsigned)(tid + 1024 * i);
            if (!((taken >> i) & 1u)) key = umax64(key, k);
        }
        return key;
    };
    unsigned long long mykey = my_best();
    for (int r = 0; r < V; ++r) {
        const unsigned long long best = block_max_u64_alt(red, mykey, lane, wave, r);
        if (best == 0ull) break;
        const float bv = key_f32((unsigned)(best >> 32));
        const int bi = (int)(unsigned)best;
        if (r >= topk && bv != vk) break;
        const float cr = (float)(tot - cum_before);
        if (cr <= d->cfg.top_p_threshold && r >= d->cfg.min_keep) break;
        if (tid == (bi & 1023)) { taken |= 1u << (bi >> 10); kept |= 1u << (bi >> 10); mykey = my_best(); }
        if (r == topk - 1) vk = bv;
        cum_before += (double)(expf(bv - mx) * inv);
    }
    }
    __syncthreads();                                                 // the last round's slot reads are done before `red` is reused
    if (step < d->cfg.min_new && tid == (d->cfg.eos & 1023)) kept &= ~(1u << (d->cfg.eos >> 10));      // gpt.py:477-478
    float m2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < TVPT; ++i) if ((kept >> i) & 1u) m2 = fmaxf(m2, x[i]);
    m2 = block_max_f(red, m2, lane, wave);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TVPT; ++i) s2 += ((kept >> i) & 1u) ? expf(x[i] - m2) : 0.f;
    s2 = block_sum_f(red, s2, lane, wave);
    const float inv2 = 1.0f / s2;
    // elements outside the kept set have p == 0 -> ratio 0 whatever q is: element 0 stands in for all of them (smallest index
    // wins ties) and only the kept ones draw noise
    unsigned long long bkey = ((unsigned long long)f32_key(0.f) << 32) | (unsigned)0x7FFFFFFF;
    if (kept != 0u) {
#pragma unroll
        for (int i = 0; i < TVPT; ++i) {
            if (!((kept >> i) & 1u)) continue;
            const int j = tid + 1024 * i;
            float qq;
            if (q != nullptr) qq = q[j];
            else {
                // same keying as the code sampler (utterance id, own step, own attempt); stream 4 marks the text row
                qq = device_exp_noise(d->seed, rs_in.uid_lo, rs_in.uid_hi, 4u, (unsigned)step, (unsigned)rs_in.attempt, j);
            }
            const float e = expf(x[i] - m2);
            bkey = umax64(bkey, ((unsigned long long)f32_key(__fdiv_rn(e * inv2, qq)) << 32) | (unsigned)(0x7FFFFFFF - j));
        }
    }
    bkey = block_max_u64(red, bkey, lane, wave);
    const int idx = 0x7FFFFFFF - (int)(unsigned)bkey;
    // next input = emb_text[idx]  (gpt.py:400-401); ids buffer keeps the reference's [.., num_vq] layout (gpt.py:492-494)
    for (int k = tid; k < a.H; k += 1024) a.x_next[(size_t)b * a.H + k] = a.emb_code[(size_t)idx * a.H + k];
    if (tid < CTTS_NUM_VQ && rs_in.fin == 0) d->ids[((size_t)seq * d->cfg.max_new + step) * CTTS_NUM_VQ + tid] = idx;      // (a finished row's tokens are never read: gpt.py:295-297)
    if (tid == 0) {
        const bool was = rs_in.fin != 0;
        const bool eos = (rs_in.fin & 2) != 0 || (idx == d->cfg.eos);                // gpt.py:490-491
        bool fin = was || eos;
        const int end_out = fin ? rs_in.end : rs_in.end + 1;                          // gpt.py:530-531
        if (!fin) d->end_idx[seq] = end_out;
        fin = fin || (end_out >= rs_in.limit);                                        // the row's own token limit (<= max_new_token, the loop bound of gpt.py:389)
        if (!was) d->finish[seq] = eos ? 1 : 0;
        ((int2*)(a.finend + b))[0] = make_int2(fin ? (eos ? 3 : 1) : 0, end_out);
        RowMeta m = a.meta[b];
        if (!was) { m.pos += 1; m.slot += 1; a.meta[b] = m; }                         // a finished row stays on its last slot (like the code sampler's rows)
        pos_s = m.pos;
        const int add = 1 + ((fin && !was) ? 0x10000 : 0);
        const int tot_t = __hip_atomic_fetch_add(&st->ticket, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
        if ((tot_t & 0xFFFF) == a.B) {
            const int nfin = tot_t >> 16;
            st->ticket = nfin << 16;
            st->step = gstep + 1;
            st->draw = draw + 1;
            if (nfin == a.B) st->all_done = 1;                                        // (every row ends by its limit <= max_new_token)
        }
    }
    __syncthreads();
    if (tid < 64) a.rope_rows[(size_t)b * 64 + tid] = a.rope[(size_t)pos_s * 64 + tid];
}

// stand-alone mode (ctts_sampler_run): block = 4 rows
__global__ __launch_bounds__(256) void sampler_rows_kernel(const SamplerArgs a) {
    const int rows = a.B;
    const SamplerDynPtr d = (SamplerDynPtr)a.dyn;
    __shared__ float tab[17];
    __shared__ unsigned long long cand_s[4][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < 17) tab[tid] = d->cfg.penalty_table[tid];
    __syncthreads();
    const int row = blockIdx.x * 4 + w;
    if (row >= rows) return;
    RowIn in;
    const float* logits = a.logits + (size_t)row * a.V;
    in.q = d->noise + (size_t)row * a.V;
    const int nh = min(a.hist_len, d->cfg.past_window);              // the last nh ids of the row's history (processors.py:21-23)
    in.myid = (lane < nh) ? a.history[(size_t)row * a.hist_len + (a.hist_len - nh) + lane] : -1;
    in.T = d->cfg.temperature[row % CTTS_NUM_VQ];
    in.penalize = d->cfg.use_penalty && (row < d->cfg.max_input_ids) && nh > 0;
    in.step = a.step_override;
    in.seed = 0; in.uid_lo = (unsigned)row; in.uid_hi = 0; in.vq = 0; in.attempt = 0;      // (stand-alone mode always receives q)
    float lg[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { const int j = lane + 64 * i; lg[i] = (j < a.V) ? logits[j] : 0.f; }
    const int idx = sample_row(knobs_of(d), tab, in, a.V, lane, lg, cand_s[w]);
    if (lane == 0) a.idx_out[row] = idx;
}

// test hook (ctts_sampler_noise): the device noise of one multinomial row, through the very function the samplers call
__global__ void noise_probe_kernel(RowIn in, int n, float* out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = exp_noise_of(in, j);
}
extern "C" int ctts_sampler_noise(uint64_t seed, uint64_t utt_id, int stream_id, int step, int attempt, int n, float* out_dev, void* stream) {
    if (!out_dev || n < 1 || stream_id < 0 || stream_id > 4 || step < 0 || attempt < 0) { ctts_set_error("sampler_noise: bad argument"); return 1; }
    RowIn in = {};
    in.q = nullptr; in.step = step; in.seed = seed; in.uid_lo = (unsigned)utt_id; in.uid_hi = (unsigned)(utt_id >> 32); in.vq = (unsigned)stream_id; in.attempt = (unsigned)attempt;
    hipLaunchKernelGGL(noise_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, n, out_dev);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_sampler(const SamplerArgs& a, int blocks, hipStream_t s) {
    if (!a.text_mode && a.V > 64 * VPL) { ctts_set_error("sampler: vocab %d > %d", a.V, 64 * VPL); return 1; }
    if (a.st != nullptr && a.text_mode) {
        if (a.V > 1024 * TVPT) { ctts_set_error("text sampler: vocab %d > %d", a.V, 1024 * TVPT); return 1; }
        hipLaunchKernelGGL(sampler_text_kernel, dim3(a.B), dim3(1024), 0, s, a);
    } else if (a.st != nullptr) hipLaunchKernelGGL(sampler_generate_kernel, dim3(a.B), dim3(256), 0, s, (const int*)a.st, a.logits, a.dyn, (const RowMeta*)a.meta, a.V, (const int*)a.hist_ring, (const RowState*)a.finend, a);
    else hipLaunchKernelGGL(sampler_rows_kernel, dim3(blocks), dim3(256), 0, s, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- small helper kernels --------------------------------------------------------------------

// ensure_non_empty regenerate (gpt.py:496-525): every row starts over; a row that itself ended at step 0 moves on to its next noise
// attempt, the others keep theirs (device noise: they re-draw the very token they drew before -- a row's result does not depend on
// which other rows shared its batch)
__global__ void restart_rows_kernel(RowState* rows, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    RowState r = rows[b];
    r.attempt += (r.fin & 2) ? 1 : 0;
    r.fin = 0; r.end = 0;
    rows[b] = r;
}
int launch_restart_rows(RowState* rows, int B, hipStream_t s) {
    hipLaunchKernelGGL(restart_rows_kernel, dim3((B + 63) / 64), dim3(64), 0, s, rows, B);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// Finished-row compaction (ctts_gpt_compact): per-row decode state of the kept rows -> rows 0..n_keep-1.  Two launches (gather into
// the c* buffers, copy back): a kept row's new place can be another kept row's old one.  The second one also re-derives the step
// state that depends on the batch: rows in the batch, how many of them have already finished, all-finished flag.
__global__ __launch_bounds__(256) void compact_gather_kernel(const int* keep, int H, const float* x, const float* rope_rows, const RowMeta* meta, const int* ring,
                                                           const RowState* fin, float* cx, float* crope, RowMeta* cmeta, int* cring, RowState* cfin) {
    const int r = blockIdx.x, src = keep[r], tid = threadIdx.x;
    for (int k = tid; k < H; k += 256) cx[(size_t)r * H + k] = x[(size_t)src * H + k];
    if (tid < 64) { crope[r * 64 + tid] = rope_rows[src * 64 + tid]; cring[r * 64 + tid] = ring[src * 64 + tid]; }
    if (tid == 0) { cmeta[r] = meta[src]; cfin[r] = fin[src]; }
}
__global__ __launch_bounds__(256) void compact_scatter_kernel(int n_keep, int H, float* x, float* rope_rows, RowMeta* meta, int* ring, RowState* fin,
                                                            const float* cx, const float* crope, const RowMeta* cmeta, const int* cring, const RowState* cfin, DevState* st) {
    const int r = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < H; k += 256) x[(size_t)r * H + k] = cx[(size_t)r * H + k];
    if (tid < 64) { rope_rows[r * 64 + tid] = crope[r * 64 + tid]; ring[r * 64 + tid] = cring[r * 64 + tid]; }
    if (tid == 0) { meta[r] = cmeta[r]; fin[r] = cfin[r]; }
    if (r == 0 && tid == 0) {
        int nfin = 0;
        for (int i = 0; i < n_keep; ++i) nfin += cfin[i].fin ? 1 : 0;
        st->ticket = nfin << 16;                      // sampler: arrivals (low half) | finished rows of the batch (high half)
        st->B = n_keep;
        if (nfin == n_keep) st->all_done = 1;
    }
}
int launch_compact_rows(const int* keep, int n_keep, int H, float* x, float* rope_rows, RowMeta* meta, int* ring, RowState* fin,
                        float* cx, float* crope, RowMeta* cmeta, int* cring, RowState* cfin, DevState* st, hipStream_t s) {
    hipLaunchKernelGGL(compact_gather_kernel, dim3(n_keep), dim3(256), 0, s, keep, H, (const float*)x, (const float*)rope_rows, (const RowMeta*)meta, (const int*)ring,
                       (const RowState*)fin, cx, crope, cmeta, cring, cfin);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(n_keep), dim3(256), 0, s, n_keep, H, x, rope_rows, meta, ring, fin, (const float*)cx, (const float*)crope,
                       (const RowMeta*)cmeta, (const int*)cring, (const RowState*)cfin, st);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}


// ctts_gpt_admit: one wavefront per new utterance.  Prompt rows (all tokens but the last) get the metadata of an ordinary prompt pass into
// KV lane seqs[i], from slot 0; the decode row rows[i] is re-initialised: input = the last prompt token's embedding at (slot T-1,
// pos cumsum-1), empty penalty window, fresh RowState, finish / end_idx of the utterance cleared.  The rows being replaced are finished
// ones: the batch's finished-row count goes down by n.
__global__ __launch_bounds__(64) void admit_rows_kernel(const AdmitArgs a) {
    const int i = blockIdx.x, lane = threadIdx.x, T = a.T;
    const int row = a.rows[i], seq = a.seqs[i];
    int cum = 0, pad = 0;
    bool seen = false;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const int mk = (t < T) ? (a.mask[i * T + t] != 0) : 0;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(mk != 0);
        if (!seen) {
            if (bal != 0ull) { pad = t0 + (int)__builtin_ctzll(bal); seen = true; }
            else pad = min(t0 + 64, T);
        }
        const int incl = cum + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u)) + mk;
        RowMeta m;
        m.seq = seq;
        m.pos = mk ? incl - 1 : 1;             // gpt.py:238-245
        m.slot = t;
        m.kv_start = mk ? pad : t;
        if (t < T - 1) a.pm[i * (T - 1) + t] = m;
        for (int j = 0; j < 64 && t0 + j < T - 1; ++j) {          // the prompt rows' RoPE table rows, one coalesced 256-byte copy each
            const int pj = __shfl(m.pos, j);
            a.rope_pre[((size_t)i * (T - 1) + t0 + j) * 64 + lane] = a.rope[(size_t)pj * 64 + lane];
        }
        cum += __builtin_popcountll(bal);
    }
    const int pos_last = max(cum - 1, 0);
    a.rope_dec[(size_t)row * 64 + lane] = a.rope[(size_t)pos_last * 64 + lane];
    a.ring[(size_t)row * 64 + lane] = -1;
    for (int k = lane; k < a.H; k += 64) a.x_dec[(size_t)row * a.H + k] = a.emb[((size_t)i * T + T - 1) * a.H + k];
    if (lane == 0) {
        RowMeta d;
        d.seq = seq; d.pos = pos_last; d.slot = T - 1; d.kv_start = pad;
        a.dm[row] = d;
        const RowState r = a.fresh[i];
        a.finend[row] = r;
        a.finish[r.out] = 0; a.end_idx[r.out] = 0;
        if (i == 0) { a.st->ticket -= a.n << 16; a.st->all_done = 0; }
    }
}
int launch_admit_rows(const AdmitArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(admit_rows_kernel, dim3(a.n), dim3(64), 0, s, a);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// rows (b, T-1) of the prompt that live in the pass [r0, r0 + n) -> dst[b]: the row indices are computed on the device, so the
// prompt pass needs no host-side index table (and no stream synchronisation between passes)
__global__ void gather_last_rows_kernel(const float* src, float* dst, int T, int r0, int n, int H) {
    const int b = blockIdx.x;
    const int sr = b * T + T - 1 - r0;
    if (sr < 0 || sr >= n) return;             // row not part of this prefill pass
    for (int k = threadIdx.x; k < H; k += blockDim.x) dst[(size_t)b * H + k] = src[(size_t)sr * H + k];
}
int launch_gather_last_rows(const float* src, float* dst, int B, int T, int r0, int n, int H, hipStream_t s) {
    hipLaunchKernelGGL(gather_last_rows_kernel, dim3(B), dim3(256), 0, s, src, dst, T, r0, n, H);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// teacher forcing: x[b] = sum_vq emb_code[vq][ids[b][vq]]  (gpt.py:403-407)
__global__ void embed_ids_kernel(const int* ids, const float* emb_code, float* x, int V, int H) {
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        float s = emb_code[((size_t)0 * V + ids[b * CTTS_NUM_VQ]) * H + k];
        for (int v = 1; v < CTTS_NUM_VQ; ++v) s += emb_code[((size_t)v * V + ids[b * CTTS_NUM_VQ + v]) * H + k];
        x[(size_t)b * H + k] = s;
    }
}
int launch_embed_ids(const int* ids, const float* emb_code, float* x, int B, int V, int H, hipStream_t s) {
    hipLaunchKernelGGL(embed_ids_kernel, dim3(B), dim3(256), 0, s, ids, emb_code, x, V, H);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// position ids / cache slots from the left-padded attention mask (gpt.py:238-245):
//   pos = cumsum(mask) - 1, pad -> 1;   decode rows start at slot T with pos = (#valid tokens)
__global__ __launch_bounds__(64) void fill_meta_kernel(RowMeta* pm, RowMeta* dm, DevState* st, const int* mask, int B, int T, const float* rope, float* rope_pre) {
    // one wavefront per sequence: 64 positions per round, cumsum(mask) from ballots (the former single-thread loop over T took
    // 0.29 ms at T = 512 -- 2 % of a 32 x 512-token prompt pass)
    const int b = blockIdx.x, lane = threadIdx.x;
    int cum = 0, pad = 0;                      // uniform: valid tokens before this round; leading zeros (left padding)
    bool seen = false;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const int mk = (t < T) ? (mask[b * T + t] != 0) : 0;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(mk != 0);
        if (!seen) {                           // first attended position so far
            if (bal != 0ull) { pad = t0 + (int)__builtin_ctzll(bal); seen = true; }
            else pad = min(t0 + 64, T);
        }
        const int incl = cum + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u)) + mk;   // cumsum(mask)[t]
        RowMeta m;
        m.seq = b;
        m.pos = mk ? incl - 1 : 1;             // gpt.py:238-245: cumsum - 1, pad -> 1
        m.slot = t;
        m.kv_start = mk ? pad : t;             // pad query rows attend to themselves only (their output is never used)
        if (t < T) pm[b * T + t] = m;
        cum += __builtin_popcountll(bal);
    }
    if (lane == 0) {
        RowMeta d;
        d.seq = b; d.pos = cum - 1; d.slot = T - 1; d.kv_start = pad;   // every sample phase advances pos/slot by one
        dm[b] = d;
        st->pad[b] = pad;
        if (b == 0) { st->step = 0; st->all_done = 0; st->ticket = 0; st->B = B; st->T = T; }
    }
}
// per prompt row: its RoPE table row (cos | sin of its position), one wavefront per row, coalesced 256-byte copies
__global__ __launch_bounds__(256) void rope_rows_kernel(const RowMeta* pm, const float* rope, float* rope_pre, int n) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r < n) rope_pre[(size_t)r * 64 + lane] = rope[(size_t)pm[r].pos * 64 + lane];
}
int launch_fill_meta(RowMeta* pm, RowMeta* dm, DevState* st, const int* mask, int B, int T, const float* rope, float* rope_pre, hipStream_t s) {
    hipLaunchKernelGGL(fill_meta_kernel, dim3(B), dim3(64), 0, s, pm, dm, st, mask, B, T, rope, rope_pre);
    hipLaunchKernelGGL(rope_rows_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, (const RowMeta*)pm, rope, rope_pre, B * T);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}

// GPT.forward "get_emb" (gpt.py:125-149) + Tokenizer.apply_spk_emb (tokenizer.py:150-178) for the prompt rows:
//   text row  -> emb_text[id0];  code row -> ((e0[id0] + e1[id1]) + e2[id2]) + e3[id3];
//   rows whose first id == spk_id <- the (already L2-normalised) speaker vector of their sequence.
__global__ void embed_prompt_kernel(const int* ids, const int* text_mask, const float* emb_text, const float* emb_code, const float* spk,
                                    int spk_id, float* out, int T, int V, int H) {
    const int row = blockIdx.x, b = row / T;
    const int* id = ids + (size_t)row * CTTS_NUM_VQ;
    const bool is_spk = (spk != nullptr) && (id[0] == spk_id);
    const bool is_text = text_mask[row] != 0;
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        float v;
        if (is_spk) v = spk[(size_t)b * H + k];
        else if (is_text) v = emb_text[(size_t)id[0] * H + k];
        else {
            v = emb_code[((size_t)0 * V + id[0]) * H + k];
            for (int q = 1; q < CTTS_NUM_VQ; ++q) v += emb_code[((size_t)q * V + id[q]) * H + k];
        }
        out[(size_t)row * H + k] = v;
    }
}
int launch_embed_prompt(const int* ids, const int* text_mask, const float* emb_text, const float* emb_code, const float* spk, int spk_id,
                        float* out, int rows, int T, int V, int H, hipStream_t s) {
    hipLaunchKernelGGL(embed_prompt_kernel, dim3(rows), dim3(256), 0, s, ids, text_mask, emb_text, emb_code, spk, spk_id, out, T, V, H);
    CTTS_HIP_CHECK(hipGetLastError());
    return 0;
}
