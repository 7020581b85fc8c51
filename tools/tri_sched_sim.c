// tri_sched_sim.c -- what a persistent, trip-synchronous wave scheduler makes of the reference scene's paths.
// Input: the per-pixel step sequences the oracle records (tools/tri_sched_sim.py writes them to /tmp), a pixel order,
// a policy.  Every wave holds 64 lanes; a lane's path is a string of steps (n/N node, T triangle, I instance, R/S ray
// complete); per trip the wave executes one or more BLOCKS (NODE, TRI, INST, SHADE), each advancing the lanes waiting for
// it by one step and costing its instruction count.  Waves run at equal rates (event-driven on accumulated cost).
// Output: total cost (VALU-bound frame time), the time the last wave ends, lane utilisation per block.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { B_NODE, B_TRI, B_INST, B_R, B_S, NB };
static int block_of(uint8_t c) { return c == 'n' || c == 'N' ? B_NODE : c == 'T' ? B_TRI : c == 'I' ? B_INST : c == 'R' ? B_R : B_S; }

typedef struct { uint64_t t; int w; } ev;
static ev* heap; static int hn;
static void push(ev e) { int i = hn++; while (i > 0 && heap[(i - 1) / 2].t > e.t) { heap[i] = heap[(i - 1) / 2]; i = (i - 1) / 2; } heap[i] = e; }
static ev pop(void) { ev top = heap[0], last = heap[--hn]; int i = 0; for (;;) { int c = 2 * i + 1; if (c >= hn) break; if (c + 1 < hn && heap[c + 1].t < heap[c].t) ++c; if (heap[c].t >= last.t) break; heap[i] = heap[c]; i = c; } heap[i] = last; return top; }

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: sim codes offs order policy waves [shade_thresh] [costs n t i r s ovh]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t nc = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* codes = malloc(nc); if (fread(codes, 1, nc, f) != nc) return 1; fclose(f);
    f = fopen(argv[2], "rb"); fseek(f, 0, SEEK_END); size_t no = ftell(f) / 8; fseek(f, 0, SEEK_SET);
    uint64_t* offs = malloc(no * 8); if (fread(offs, 8, no, f) != no) return 1; fclose(f);
    f = fopen(argv[3], "rb"); fseek(f, 0, SEEK_END); size_t np = ftell(f) / 4; fseek(f, 0, SEEK_SET);
    uint32_t* order = malloc(np * 4); if (fread(order, 4, np, f) != np) return 1; fclose(f);
    const int policy = atoi(argv[4]);      // 0: every block that has a lane, each trip; 1: the block with most lanes (shade only from thresh lanes on, or when nothing else waits)
    const int waves = atoi(argv[5]);
    const int thresh = argc > 6 ? atoi(argv[6]) : 24;
    int cost[NB] = {55, 75, 90, 400, 450}, ovh = 15;
    if (argc > 12) { for (int k = 0; k < NB; ++k) cost[k] = atoi(argv[7 + k]); ovh = atoi(argv[12]); }
    typedef struct { uint64_t pos[64], end[64]; } wave;
    wave* W = calloc(waves, sizeof(wave));
    heap = malloc(sizeof(ev) * (waves + 1));
    size_t cursor = 0;
    for (int w = 0; w < waves; ++w) push((ev){0, w});
    uint64_t total = 0, last_end = 0, lanes_run[NB] = {0}, runs[NB] = {0}, trips = 0, t_dry = 0;
    while (hn) {
        ev e = pop();
        wave* v = &W[e.w];
        // refill idle lanes (free: counted inside the shade cost)
        for (int l = 0; l < 64; ++l)
            while (v->pos[l] == v->end[l] && cursor < np) { uint32_t p = order[cursor++]; v->pos[l] = offs[p]; v->end[l] = offs[p + 1]; if (cursor == np) t_dry = e.t; }
        int cnt[NB] = {0};
        for (int l = 0; l < 64; ++l) if (v->pos[l] < v->end[l]) ++cnt[block_of(codes[v->pos[l]])];
        int any = 0; for (int k = 0; k < NB; ++k) any += cnt[k];
        if (!any) { if (e.t > last_end) last_end = e.t; continue; }
        int run[NB] = {0};
        if (policy == 0) { for (int k = 0; k < NB; ++k) run[k] = cnt[k] > 0; }
        else {
            // walk blocks by count; the shade blocks only from `thresh` waiting lanes on, or when no walk block has a lane
            int best = -1;
            for (int k = 0; k < 3; ++k) if (cnt[k] > 0 && (best < 0 || cnt[k] > cnt[best])) best = k;
            const int sh = cnt[B_R] + cnt[B_S];
            if (best < 0 || sh >= thresh) { run[B_R] = cnt[B_R] > 0; run[B_S] = cnt[B_S] > 0; }
            else run[best] = 1;
            if (policy == 2 && best >= 0 && !(run[B_R] || run[B_S])) {   // 2: also any other walk block with at least thresh2 lanes
                for (int k = 0; k < 3; ++k) if (cnt[k] >= 16) run[k] = 1;
            }
        }
        uint64_t c = ovh;
        for (int k = 0; k < NB; ++k) if (run[k]) { c += cost[k]; lanes_run[k] += cnt[k]; ++runs[k]; }
        for (int l = 0; l < 64; ++l) if (v->pos[l] < v->end[l] && run[block_of(codes[v->pos[l]])]) ++v->pos[l];
        total += c; ++trips;
        push((ev){e.t + c, e.w});
    }
    printf("policy %d waves %d thresh %d: total %.3e instr-units, per wave %.0f, last wave ends %llu (cursor dry %llu), trips %llu\n", policy, waves, thresh,
           (double)total, (double)total / waves, (unsigned long long)last_end, (unsigned long long)t_dry, (unsigned long long)trips);
    const char* nm[NB] = {"NODE", "TRI", "INST", "R", "S"};
    for (int k = 0; k < NB; ++k) printf("  %-4s runs %9llu lanes/run %.1f\n", nm[k], (unsigned long long)runs[k], runs[k] ? (double)lanes_run[k] / runs[k] : 0.0);
    return 0;
}
