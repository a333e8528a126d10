/*
 * The flowgraph signal -> demod -> decode driven through the streaming C-ABI alone (include/gr80211_host.h), in
 * plain C: what a block wrapper in another language does, without pmt.  Three blocks joined by item buffers; tags
 * travel as gr80211_tag records, vector values copied because a block only keeps them until its next call.
 *
 *   gr80211_abi_chain <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]
 *       triggers.txt: one "pos rad snr rssi" line per tagged sync byte; iq*.bin: interleaved float32 I/Q
 *
 * Output: "PDU <hex of the message blob>" per message, then "STAT pdus=N".  Exit code 0 on success.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gr80211_host.h"

#define TAGS_PER_CALL 4096

/* A stream between two blocks: items not yet consumed and the tags on them. */
typedef struct {
    size_t item;     /* bytes per item */
    char *buf;       /* items [base, base + len/item) start at buf + head */
    size_t len, cap, head;
    uint64_t base;   /* absolute index of the first unread item */
    uint64_t written; /* absolute index of the next item the producer writes */
    gr80211_tag *tags;
    size_t n_tags, cap_tags, tag_head;
} link_t;

static void die(const char *what)
{
    fprintf(stderr, "%s\n", what);
    exit(1);
}

static void link_init(link_t *l, size_t item)
{
    memset(l, 0, sizeof *l);
    l->item = item;
}

static size_t link_items(const link_t *l) { return (l->len - l->head) / l->item; }

static void link_push_items(link_t *l, const void *src, size_t n)
{
    const size_t bytes = n * l->item;
    if (!bytes) return;
    if (l->len + bytes > l->cap) {
        if (l->head) { /* drop what has been read before growing */
            memmove(l->buf, l->buf + l->head, l->len - l->head);
            l->len -= l->head;
            l->head = 0;
        }
        if (l->len + bytes > l->cap) {
            l->cap = (l->len + bytes) * 2;
            l->buf = (char *)realloc(l->buf, l->cap);
            if (!l->buf) die("out of memory");
        }
    }
    memcpy(l->buf + l->len, src, bytes);
    l->len += bytes;
    l->written += n;
}

static void link_push_tag(link_t *l, const gr80211_tag *t)
{
    if (l->n_tags == l->cap_tags) {
        l->cap_tags = l->cap_tags ? l->cap_tags * 2 : 256;
        l->tags = (gr80211_tag *)realloc(l->tags, l->cap_tags * sizeof *l->tags);
        if (!l->tags) die("out of memory");
    }
    gr80211_tag c = *t;
    if (t->cvec && t->cvec_len) {
        float *v = (float *)malloc((size_t)t->cvec_len * 2 * sizeof(float));
        if (!v) die("out of memory");
        memcpy(v, t->cvec, (size_t)t->cvec_len * 2 * sizeof(float));
        c.cvec = v;
    }
    l->tags[l->n_tags++] = c;
}

static void link_consume(link_t *l, size_t n)
{
    l->head += n * l->item;
    l->base += n;
    while (l->tag_head < l->n_tags && l->tags[l->tag_head].offset < l->base) {
        free((void *)l->tags[l->tag_head].cvec);
        l->tags[l->tag_head].cvec = NULL;
        l->tag_head++;
    }
}

static void link_free(link_t *l)
{
    for (size_t k = l->tag_head; k < l->n_tags; k++) free((void *)l->tags[k].cvec);
    free(l->tags);
    free(l->buf);
}

/* One general_work call: `chunk` items of room on every output, up to `chunk` items offered on every input. */
static int step(gr80211_block *b, link_t **in, int n_in, link_t **out, int n_out, int chunk, int *progress)
{
    size_t avail = n_in ? link_items(in[0]) : 0;
    for (int p = 1; p < n_in; p++)
        if (link_items(in[p]) < avail) avail = link_items(in[p]);
    const int win = (int)(avail < (size_t)chunk ? avail : (size_t)chunk);
    int nin[3];
    const void *ip[3];
    static gr80211_tag tin[TAGS_PER_CALL], tout[TAGS_PER_CALL];
    int n_tin = 0;
    for (int p = 0; p < n_in; p++) {
        nin[p] = win;
        ip[p] = in[p]->buf + in[p]->head;
        for (size_t k = in[p]->tag_head; k < in[p]->n_tags && in[p]->tags[k].offset < in[p]->base + (uint64_t)win; k++) {
            if (n_tin == TAGS_PER_CALL) die("too many tags in one window");
            tin[n_tin] = in[p]->tags[k];
            tin[n_tin++].port = (uint32_t)p;
        }
    }
    void *op[2] = {NULL, NULL};
    for (int p = 0; p < n_out; p++) {
        op[p] = malloc((size_t)chunk * out[p]->item);
        if (!op[p]) die("out of memory");
    }
    int consumed = 0, n_tout = 0, produced;
    if (n_out)
        produced = gr80211_block_work_out(b, chunk, nin, ip, n_in, tin, n_tin, op, n_out, tout, TAGS_PER_CALL, &n_tout,
                                          &consumed);
    else
        produced = gr80211_block_work(b, chunk, nin, ip, n_in, tin, n_tin, &consumed);
    if (produced < 0) {
        fprintf(stderr, "block error: %s\n", gr80211_block_last_error(b));
        for (int p = 0; p < n_out; p++) free(op[p]);
        return 0;
    }
    for (int k = 0; k < n_tout; k++) {
        if ((int)tout[k].port >= n_out) die("tag on an output port that does not exist");
        link_push_tag(out[tout[k].port], &tout[k]);
    }
    for (int p = 0; p < n_out; p++) {
        link_push_items(out[p], op[p], (size_t)produced);
        free(op[p]);
    }
    for (int p = 0; p < n_in; p++) link_consume(in[p], (size_t)consumed);
    if (consumed > 0 || produced > 0) *progress = 1;
    return 1;
}

static unsigned long long g_pdus = 0;

static void on_msg(void *user, const uint8_t *blob, int len)
{
    (void)user;
    g_pdus++;
    fputs("PDU ", stdout);
    for (int i = 0; i < len; i++) printf("%02x", blob[i]);
    fputc('\n', stdout);
}

static char *read_file(const char *path, size_t *bytes)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *p = (char *)malloc(n > 0 ? (size_t)n : 1);
    if (p && n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) {
        free(p);
        p = NULL;
    }
    fclose(f);
    *bytes = (size_t)n;
    return p;
}

int main(int argc, char **argv)
{
    if (argc < 6) {
        fprintf(stderr, "usage: gr80211_abi_chain <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]\n");
        return 2;
    }
    const int n_rx = atoi(argv[1]), chunk = atoi(argv[2]), batch = atoi(argv[3]);
    if ((n_rx != 1 && n_rx != 2) || chunk < 1 || (n_rx == 2 && argc < 7)) return 2;
    size_t b0 = 0, b1 = 0;
    char *iq0 = read_file(argv[5], &b0), *iq1 = n_rx == 2 ? read_file(argv[6], &b1) : NULL;
    if (!iq0 || (n_rx == 2 && !iq1)) die("cannot read IQ file");
    size_t n = b0 / 8;
    if (n_rx == 2 && b1 / 8 < n) n = b1 / 8;

    gr80211_block *sg = gr80211_block_create(n_rx == 2 ? GR80211_BLOCK_SIGNAL2 : GR80211_BLOCK_SIGNAL, batch, 0);
    gr80211_block *dm = gr80211_block_create(n_rx == 2 ? GR80211_BLOCK_DEMOD2 : GR80211_BLOCK_DEMOD, batch, 0);
    gr80211_block *dc = gr80211_block_create(GR80211_BLOCK_DECODE, batch, 0);
    if (!sg || !dm || !dc) {
        fprintf(stderr, "create failed: %s\n", gr80211_block_last_error(NULL));
        return 1;
    }
    if (gr80211_block_n_inputs(sg) != n_rx + 1 || gr80211_block_n_outputs(sg) != n_rx ||
        gr80211_block_n_inputs(dm) != n_rx || gr80211_block_n_outputs(dm) != 1 || gr80211_block_n_outputs(dc) != 0)
        die("unexpected port counts");
    gr80211_block_set_msg_handler(dc, on_msg, NULL);

    link_t l_sync, l_a, l_b, l_s0, l_s1, l_llr;
    link_init(&l_sync, 1);
    link_init(&l_a, 8);
    link_init(&l_b, 8);
    link_init(&l_s0, 8);
    link_init(&l_s1, 8);
    link_init(&l_llr, 4);
    char *sync = (char *)calloc(n ? n : 1, 1);
    if (!sync) die("out of memory");
    FILE *f = fopen(argv[4], "r");
    if (!f) die("cannot read the trigger file");
    unsigned long long pos, last = 0;
    float rad, snr, rssi;
    while (fscanf(f, "%llu %f %f %f", &pos, &rad, &snr, &rssi) == 4) {
        if (pos >= n) continue;
        if (pos < last) die("triggers must come in stream order");
        last = pos;
        sync[pos] = 1;
        gr80211_tag t;
        memset(&t, 0, sizeof t);
        t.offset = pos;
        t.key = GR80211_TAG_RAD; t.value = rad; link_push_tag(&l_sync, &t);
        t.key = GR80211_TAG_SNR; t.value = snr; link_push_tag(&l_sync, &t);
        t.key = GR80211_TAG_RSSI; t.value = rssi; link_push_tag(&l_sync, &t);
    }
    fclose(f);
    link_push_items(&l_sync, sync, n);
    link_push_items(&l_a, iq0, n);
    if (n_rx == 2) link_push_items(&l_b, iq1, n);
    free(sync);
    free(iq0);
    free(iq1);

    link_t *in_sg[3] = {&l_sync, &l_a, &l_b}, *out_sg[2] = {&l_s0, &l_s1}, *in_dm[2] = {&l_s0, &l_s1};
    link_t *out_dm[1] = {&l_llr}, *in_dc[1] = {&l_llr};
    int ok = 1, flushed_sg = 0, flushed_dm = 0;
    for (int idle = 0; ok && idle < 4;) {
        int progress = 0;
        ok = step(sg, in_sg, n_rx + 1, out_sg, n_rx, chunk, &progress) &&
             step(dm, in_dm, n_rx, out_dm, 1, chunk, &progress) && step(dc, in_dc, 1, NULL, 0, chunk, &progress);
        if (progress) {
            idle = 0;
            continue;
        }
        /* nothing moves any more: end of stream for the next block that still holds frames back */
        if (!flushed_sg) { ok = gr80211_block_flush(sg) == 0; flushed_sg = 1; continue; }
        if (!flushed_dm) { ok = gr80211_block_flush(dm) == 0; flushed_dm = 1; continue; }
        idle++;
    }
    ok = ok && gr80211_block_flush(dc) == 0;
    if (gr80211_block_pending_output(sg) != 0 || gr80211_block_pending_output(dm) != 0) ok = 0;
    printf("STAT pdus=%llu\n", g_pdus);
    gr80211_block_destroy(sg);
    gr80211_block_destroy(dm);
    gr80211_block_destroy(dc);
    link_free(&l_sync);
    link_free(&l_a);
    link_free(&l_b);
    link_free(&l_s0);
    link_free(&l_s1);
    link_free(&l_llr);
    return ok ? 0 : 1;
}
