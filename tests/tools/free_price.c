/*
 * free_price.c -- TEST INFRASTRUCTURE (pricing tool; it includes the CPU oracle's source, hence it lives under tests/).
 *
 * Prices a FREE-RUNNING search for move-steps with many live games before it is built (round-5 review, item 4): today the
 * engine runs one network launch per MCTS iteration while more than 256 games live -- 300 us for 300 boards, 468 us for 600,
 * 577 us for 1024: the launch is sized by the live games and the chip is part empty.  The reference couples the games of a
 * call only through `node_selected` (alpha_mcts.rs:151,170) and the stale slots (:142,192-200), so each game may run its own
 * iteration counter: per ROUND one full-size launch of R rows carries every game's demanded leaf plus speculative rows (the
 * demanded leaf's children, created ahead with the dice of the iteration that will expand it; the unexpanded nodes that virtual
 * PUCT descents end on -- the tail's policy, die-e_amd/csrc/mcts_kernels.hip k_tail), and each game then runs iterations for as
 * long as its selected leaf is a finished game or has its evaluation.  This file simulates exactly that on the oracle's rules
 * and tree arithmetic (quirks off: a game never waits for another) and counts the rounds a move-step's search needs.
 * lockstep = 1 simulates today's k_tail instead (an iteration runs only when EVERY game's leaf is at hand): the calibration
 * against the measured iterations per launch (profiles/r05H_*, r05M_*).
 */
#include "../../oracle/diee_oracle.c"

typedef struct {
    int iterations;
    float c, dir_alpha, dir_eps;
    int rows;            /* R: rows of a launch */
    int child_rows;      /* children of the demanded leaf a game may wish for (0: none) */
    int cand_max;        /* virtual-descent candidates a game may wish for */
    int rollouts;        /* virtual descents per game and round at most */
    int order;           /* wish order behind the demanded leaf: 0 candidates then children, 1 first candidate, children, other candidates,
                            2 children first when the first virtual descent lands on the demanded leaf (the search is about to deepen), else as 1 */
    int lockstep;        /* 1: today's k_tail (all games iterate together or not at all) */
    int root_children;   /* 1: the root launch's free rows carry the roots' children (they exist before the evaluation: rules + dice) */
    int share_cap;       /* 0: free rows by rank over all games; > 0: at most this many speculative rows per game and launch */
    int preexpand;       /* 1: PRICING of round-5 review item 5 (path-keyed child dice): a node's children are created -- states from dice keyed by the node's
                            PATH, priors from its evaluation -- as soon as its evaluation arrives, invisible to the real search until it expands the node
                            (then a commit), visible to the virtual descents: they run on into evaluated nodes and name GRANDCHILDREN as candidates */
    int prio;            /* grant order of the free rows: 0 rank by rank, games in slot order; 1 rank by rank, the games furthest behind first;
                            2 the games furthest behind get ALL their wishes first */
} fp_cfg;

typedef struct {
    long launches, rows, rows_demanded, rows_used, iterations_run, game_rounds, stalls;
    long adv_hist[16];   /* iterations a game completed in a round (15 = 15 or more) */
    long rounds_of_game_max, rounds_of_game_sum;
} fp_out;

typedef struct {
    uint8_t have;        /* evaluation at hand */
    uint8_t used;
    float value;
    int k;               /* legal plays of the state */
    float* pr;           /* their policy entries P[code_j] (unnormalised), in play order */
} fp_eval_t;

typedef struct {
    or_store st;
    fp_eval_t* ev; int ev_cap;
    uint32_t gid, rnd;
    int it, leaf, lterm, done, root_player, rounds;
    /* children evaluated ahead of the expansion of `ahead_leaf` (stashed until the expansion creates the nodes) */
    int ahead_leaf, ahead_n; fp_eval_t ahead[64];
    int* wish; int n_wish, wish_children0, wish_nchildren;   /* wish[i] >= 0: node; wish[i] = -1 - j: child j of the demanded leaf */
    or_state* child_state;                                    /* [child_rows] states of the children created ahead */
    float* vvis; float* vval; int vcap;
    int* sh_first; int* sh_k; int sh_cap;                     /* preexpand: children created ahead of a node's expansion (first, count; 0: none) */
} fp_game;

static void ev_grow(fp_game* G) {
    if (G->st.n <= G->ev_cap) return;
    int nc = G->ev_cap ? G->ev_cap : 1024;
    while (nc < G->st.n) nc *= 2;
    G->ev = realloc(G->ev, sizeof(fp_eval_t) * (size_t)nc);
    memset(G->ev + G->ev_cap, 0, sizeof(fp_eval_t) * (size_t)(nc - G->ev_cap));
    G->sh_first = realloc(G->sh_first, sizeof(int) * (size_t)nc); G->sh_k = realloc(G->sh_k, sizeof(int) * (size_t)nc);
    memset(G->sh_first + G->ev_cap, 0, sizeof(int) * (size_t)(nc - G->ev_cap)); memset(G->sh_k + G->ev_cap, 0, sizeof(int) * (size_t)(nc - G->ev_cap));
    G->ev_cap = nc;
}

/* dice key of a node's children when they are keyed by the node's path from the root (round-5 review item 5) instead of the iteration that expands it */
static uint32_t path_key(const fp_game* G, int idx) {
    int ord[256], d = 0;
    for (int i = idx; G->st.nodes[i].parent >= 0 && d < 256; i = G->st.nodes[i].parent) {
        const int p = G->st.nodes[i].parent;
        int fc = G->st.nodes[p].n_children > 0 ? G->st.nodes[p].first_child : G->sh_first[p];
        ord[d++] = i - fc;
    }
    uint32_t h = 0;
    for (int k = d - 1; k >= 0; --k) { h = (h ^ ((uint32_t)ord[k] + 0x9E3779B9u)) * 0x85EBCA6Bu; h ^= h >> 13; }
    if (h >= 0xFFFFFFF0u) h -= 0x10u;
    return idx == 0 ? 0u : (h | 1u);
}

/* the evaluation of one state -> what an expansion needs of it */
static void ev_fill(const or_game* g, fp_eval_t* e, const or_state* s, const float* policy_row, float value, or_play* plays) {
    int k = g->valid_moves(s, plays, OR_MAX_PLAYS);
    e->have = 1; e->used = 0; e->value = value; e->k = k;
    e->pr = malloc(sizeof(float) * (size_t)(k > 0 ? k : 1));
    for (int j = 0; j < k; ++j) e->pr[j] = policy_row[g->encode(s, &plays[j])];
}

static int is_terminal(const or_game* g, const or_state* s, int* winner) { return g->check_winner(s, winner); }

/* alpha_expand_tensor with the priors of a stored evaluation (expand_node's arithmetic: sequential sum in play order).  shadow = 1: the children are
 * created but the node stays a leaf for the real search (preexpand); a later real expansion only commits them */
static void fp_expand(const or_game* g, fp_game* G, int idx, const fp_eval_t* e, uint64_t seed, uint32_t eit, const float* noise, float eps,
                      or_play* plays, int shadow) {
    or_node* nd = &G->st.nodes[idx];
    if (nd->drained) return;
    if (!shadow && G->sh_k[idx] > 0) {                /* commit what was created ahead */
        nd->first_child = G->sh_first[idx]; nd->n_children = G->sh_k[idx]; nd->drained = 1;
        return;
    }
    if (shadow && G->sh_k[idx] != 0) return;
    or_state s = nd->state;
    int k = g->valid_moves(&s, plays, OR_MAX_PLAYS);
    float* p = malloc(sizeof(float) * (size_t)(k > 0 ? k : 1));
    float sum = 0.0f;
    for (int j = 0; j < k; ++j) {
        float x = e->pr[j];
        if (noise) x = (1.0f - eps) * x + eps * noise[g->encode(&s, &plays[j])];
        p[j] = x; sum += x;
    }
    int first = G->st.n;
    for (int j = 0; j < k; ++j) {
        or_state ns = s;
        uint8_t d0, d1;
        or_dice(seed, G->gid, G->rnd, eit, (uint32_t)j, &d0, &d1);
        g->apply_move(&ns, &plays[j], d0, d1);
        store_add(&G->st, &ns, idx, (int)g->encode(&s, &plays[j]), p[j] / sum);
    }
    free(p);
    ev_grow(G);
    nd = &G->st.nodes[idx];
    if (shadow) { G->sh_first[idx] = first; G->sh_k[idx] = k > 0 ? k : -1; return; }      /* (-1: nothing to create; the real expansion drains the node) */
    nd->first_child = k ? first : -1; nd->n_children = k; nd->drained = 1;
    if (G->ahead_leaf == idx) {                       /* the children evaluated ahead of this expansion take their rows */
        for (int j = 0; j < G->ahead_n && j < k; ++j) G->ev[first + j] = G->ahead[j];
        for (int j = k; j < G->ahead_n; ++j) free(G->ahead[j].pr);
        G->ahead_leaf = -1; G->ahead_n = 0;
    }
}

/* selection for the game's next iteration; a finished game is backpropagated at once (alpha_mcts.rs:157-163) */
static void fp_select(const or_game* g, fp_game* G, float c) {
    int d;
    G->leaf = or_select_leaf(&G->st, 0, c, &d);
    int w;
    if (is_terminal(g, &G->st.nodes[G->leaf].state, &w)) {
        or_backpropagate(&G->st, G->leaf, w == G->root_player ? 1.0f : -1.0f);
        G->lterm = 1;
    } else G->lterm = 0;
}

/* can the game's pending iteration run now? */
static int fp_ready(const fp_game* G) { return G->lterm || G->ev[G->leaf].have; }

static void fp_iterate(const or_game* g, fp_game* G, const fp_cfg* cfg, uint64_t seed, or_play* plays, fp_out* out) {
    if (!G->lterm) {
        fp_eval_t* e = &G->ev[G->leaf];
        if (!e->used) { e->used = 1; out->rows_used++; }
        const float v = e->value;
        fp_expand(g, G, G->leaf, e, seed, cfg->preexpand ? path_key(G, G->leaf) : (uint32_t)G->it + 1u, NULL, 0.0f, plays, 0);
        or_backpropagate(&G->st, G->leaf, v);
    }
    G->it++;
    out->iterations_run++;
    if (G->it >= cfg->iterations) { G->done = 1; return; }
    fp_select(g, G, cfg->c);
}

/* the game's wishes for the next launch: [demanded leaf] then speculative rows in the order cfg->order says */
static void fp_plan(const or_game* g, fp_game* G, const fp_cfg* cfg, uint64_t seed, or_play* plays) {
    G->n_wish = 0; G->wish_nchildren = 0;
    if (G->done) return;
    const int demanded = (!G->lterm && !G->ev[G->leaf].have) ? G->leaf : -1;
    if (demanded >= 0) G->wish[G->n_wish++] = demanded;
    /* virtual descents on a scratch copy of visits / value (k_tail's plan) */
    int cand[64], ncand = 0, first_hit_demanded = 0;
    if (cfg->cand_max > 0 && cfg->rollouts > 0) {
        if (G->vcap < G->st.n) { G->vcap = G->st.n * 2; G->vvis = realloc(G->vvis, sizeof(float) * (size_t)G->vcap); G->vval = realloc(G->vval, sizeof(float) * (size_t)G->vcap); }
        for (int i = 0; i < G->st.n; ++i) { G->vvis[i] = G->st.nodes[i].visits; G->vval[i] = G->st.nodes[i].value; }
        int fruitless = 0;
        for (int step = 0; step < cfg->rollouts && ncand < cfg->cand_max && ncand < 64 && fruitless < 8; ++step) {
            int idx = 0;
            for (;;) {
                const or_node* nd = &G->st.nodes[idx];
                int nk = nd->n_children, nf = nd->first_child;
                if (nk == 0 && cfg->preexpand && G->sh_k[idx] > 0) { nk = G->sh_k[idx]; nf = G->sh_first[idx]; }      /* created ahead: the descent goes on */
                if (nk == 0) break;
                const float sq = sqrtf(G->vvis[idx]);
                int best = -1; float bs = 0.0f;
                for (int j = 0; j < nk; ++j) {
                    const int ch = nf + j;
                    const float vis = G->vvis[ch], val = G->vval[ch];
                    const float q = vis == 0.0f ? 0.0f : val / vis;
                    const float s = q + (cfg->c * (sq / (vis + 1.0f))) * G->st.nodes[ch].policy;
                    if (s == s && (best < 0 || !(bs > s))) { best = ch; bs = s; }
                }
                idx = best >= 0 ? best : nf + nk - 1;
            }
            float x = 0.0f;
            int w, fresh = 0;
            if (is_terminal(g, &G->st.nodes[idx].state, &w)) x = w == G->root_player ? 1.0f : -1.0f;
            else if (G->ev[idx].have) x = G->ev[idx].value;
            else if (!G->st.nodes[idx].drained && idx != demanded) {
                fresh = 1;
                for (int i = 0; i < ncand; ++i) if (cand[i] == idx) fresh = 0;
            }
            if (step == 0 && idx == demanded && demanded >= 0) first_hit_demanded = 1;
            if (fresh) { cand[ncand++] = idx; fruitless = 0; } else ++fruitless;
            for (int p = idx; p >= 0; p = G->st.nodes[p].parent) { G->vvis[p] += 1.0f; G->vval[p] += x; }
        }
    }
    /* the demanded leaf's children, as the expansion of iteration `it` will create them */
    int nchild = 0;
    if (cfg->child_rows > 0 && demanded >= 0 && !G->st.nodes[demanded].drained) {
        or_state s = G->st.nodes[demanded].state;
        int k = g->valid_moves(&s, plays, OR_MAX_PLAYS);
        nchild = k < cfg->child_rows ? k : cfg->child_rows;
        if (nchild > 64) nchild = 64;
        for (int j = 0; j < nchild; ++j) {
            or_state ns = s; uint8_t d0, d1;
            or_dice(seed, G->gid, G->rnd, (uint32_t)G->it + 1u, (uint32_t)j, &d0, &d1);
            g->apply_move(&ns, &plays[j], d0, d1);
            G->child_state[j] = ns;
        }
    }
    G->wish_nchildren = nchild;
    int children_first = cfg->order == 2 && first_hit_demanded;
    int ci = 0;
    if (!children_first && cfg->order != 0 && ncand > 0) G->wish[G->n_wish++] = cand[ci++];
    if (cfg->order == 0) while (ci < ncand) G->wish[G->n_wish++] = cand[ci++];
    for (int j = 0; j < nchild; ++j) {
        int w; if (is_terminal(g, &G->child_state[j], &w)) continue;       /* a finished game needs no evaluation */
        G->wish[G->n_wish++] = -1 - j;
    }
    while (ci < ncand) G->wish[G->n_wish++] = cand[ci++];
}

typedef void (*fp_eval_fn)(void* ctx, const or_state* states, int n, float* policy, float* value);

/* one move-step's search for m games; returns 0 */
int fp_run(const or_state* roots, int m, const fp_cfg* cfg, uint64_t seed, uint32_t step, const uint32_t* gids, const uint32_t* rounds,
           fp_eval_fn eval, void* ectx, fp_out* out) {
    const or_game* g = or_game_by_id(1);
    const int A = g->n_actions;
    memset(out, 0, sizeof *out);
    fp_game* Gs = calloc((size_t)m, sizeof(fp_game));
    or_play* plays = malloc(sizeof(or_play) * OR_MAX_PLAYS);
    const int R = cfg->rows;
    const int cap_rows = R > m ? R : m;
    or_state* bstate = malloc(sizeof(or_state) * (size_t)cap_rows);
    int* bgame = malloc(sizeof(int) * (size_t)cap_rows); int* bwish = malloc(sizeof(int) * (size_t)cap_rows);
    float* pol = malloc(sizeof(float) * (size_t)cap_rows * (size_t)A); float* val = malloc(sizeof(float) * (size_t)cap_rows);
    float* noise = malloc(sizeof(float) * (size_t)A);
    or_dirichlet(seed, step, cfg->dir_alpha, A, noise);
    /* ---- the root launch: m roots (+ their children when root_children) ---- */
    for (int i = 0; i < m; ++i) {
        fp_game* G = &Gs[i];
        or_store_init(&G->st); G->gid = gids[i]; G->rnd = rounds[i]; G->ahead_leaf = -1;
        store_add(&G->st, &roots[i], -1, -1, 0.0f);
        G->st.nodes[0].visits = 1.0f;                                   /* alpha_mcts.rs:123 */
        G->root_player = g->get_player(&roots[i]);
        G->wish = malloc(sizeof(int) * 256); G->child_state = malloc(sizeof(or_state) * 64);
        ev_grow(G);
        bstate[i] = roots[i];
    }
    eval(ectx, bstate, m, pol, val);
    out->launches++; out->rows += m; out->rows_demanded += m;
    for (int i = 0; i < m; ++i) {
        fp_game* G = &Gs[i];
        ev_fill(g, &G->ev[0], &roots[i], pol + (size_t)i * A, val[i], plays);
        G->ev[0].used = 1; out->rows_used++;
        fp_expand(g, G, 0, &G->ev[0], seed, 0u, noise, cfg->dir_eps, plays, 0);
    }
    if (cfg->root_children && R > m) {
        /* free rows of the root launch: the roots' children in turn (child j of every game, then j + 1 ...), as far as the rows go.
         * (In the engine these rows would ride in the SAME launch as the roots; here they are a second evaluator call, not a launch.) */
        int nb = 0;
        for (int j = 0; nb < R - m; ++j) {
            int any = 0;
            for (int i = 0; i < m && nb < R - m; ++i) {
                fp_game* G = &Gs[i];
                if (j >= G->st.nodes[0].n_children) continue;
                any = 1;
                const int ch = G->st.nodes[0].first_child + j;
                int w; if (is_terminal(g, &G->st.nodes[ch].state, &w)) continue;
                bstate[nb] = G->st.nodes[ch].state; bgame[nb] = i; bwish[nb] = ch; ++nb;
            }
            if (!any) break;
        }
        if (nb) {
            eval(ectx, bstate, nb, pol, val);
            out->rows += nb;
            for (int r = 0; r < nb; ++r) { fp_game* G = &Gs[bgame[r]]; ev_fill(g, &G->ev[bwish[r]], &bstate[r], pol + (size_t)r * A, val[r], plays); }
        }
    }
    int live = 0;
    for (int i = 0; i < m; ++i) {
        fp_game* G = &Gs[i];
        if (cfg->iterations <= 0) { G->done = 1; continue; }
        fp_select(g, G, cfg->c);
        ++live;
    }
    /* ---- rounds ---- */
    int* adv = malloc(sizeof(int) * (size_t)m);
    int* order_ = malloc(sizeof(int) * (size_t)m);
    while (live > 0) {
        /* iterations */
        memset(adv, 0, sizeof(int) * (size_t)m);
        if (cfg->lockstep) {
            for (;;) {
                int all = 1, any = 0;
                for (int i = 0; i < m; ++i) if (!Gs[i].done) { any = 1; if (!fp_ready(&Gs[i])) all = 0; }
                if (!any || !all) break;
                for (int i = 0; i < m; ++i) if (!Gs[i].done) { fp_iterate(g, &Gs[i], cfg, seed, plays, out); adv[i]++; }
            }
        } else {
            for (int i = 0; i < m; ++i) {
                fp_game* G = &Gs[i];
                while (!G->done && fp_ready(G)) { fp_iterate(g, G, cfg, seed, plays, out); adv[i]++; }
            }
        }
        live = 0;
        for (int i = 0; i < m; ++i) {
            fp_game* G = &Gs[i];
            if (adv[i] || !G->done) { out->adv_hist[adv[i] < 15 ? adv[i] : 15]++; out->game_rounds++; G->rounds++; }
            if (!G->done) ++live;
        }
        if (!live) break;
        /* plan + grant */
        int nb = 0;
        for (int i = 0; i < m; ++i) fp_plan(g, &Gs[i], cfg, seed, plays);
        for (int i = 0; i < m; ++i) {                      /* demanded leaves first */
            fp_game* G = &Gs[i];
            if (G->n_wish && G->wish[0] >= 0 && G->wish[0] == G->leaf && !G->lterm && !G->ev[G->leaf].have) {
                if (nb < cap_rows) { bstate[nb] = G->st.nodes[G->leaf].state; bgame[nb] = i; bwish[nb] = G->leaf; ++nb; }
            }
        }
        const int demanded_rows = nb;
        if (nb > R) { /* more demanded leaves than rows: the launch is sized by the games (today's rule) */ }
        /* grant order of the games: slot order, or the games furthest behind first (a launch ends the search only when the LAST game is done) */
        for (int i = 0; i < m; ++i) order_[i] = i;
        if (cfg->prio) {
            for (int i = 1; i < m; ++i) {                  /* insertion sort by progress, stable (m <= 1024: fine) */
                const int x = order_[i]; int j = i - 1;
                while (j >= 0 && Gs[order_[j]].it > Gs[x].it) { order_[j + 1] = order_[j]; --j; }
                order_[j + 1] = x;
            }
        }
        if (cfg->prio == 2) {
            for (int oi = 0; oi < m && nb < R; ++oi) {
                const int i = order_[oi];
                fp_game* G = &Gs[i];
                const int dem = (G->n_wish && G->wish[0] == G->leaf && !G->lterm && !G->ev[G->leaf].have) ? 1 : 0;
                for (int wi = dem; wi < G->n_wish && nb < R; ++wi) {
                    if (cfg->share_cap > 0 && wi - dem >= cfg->share_cap) break;
                    const int w = G->wish[wi];
                    bstate[nb] = w >= 0 ? G->st.nodes[w].state : G->child_state[-1 - w];
                    bgame[nb] = i; bwish[nb] = w; ++nb;
                }
            }
        } else
        for (int rank = 1; nb < R; ++rank) {
            int any = 0;
            for (int oi = 0; oi < m && nb < R; ++oi) {
                const int i = order_[oi];
                fp_game* G = &Gs[i];
                const int dem = (G->n_wish && G->wish[0] == G->leaf && !G->lterm && !G->ev[G->leaf].have) ? 1 : 0;
                const int wi = rank - 1 + dem;
                if (wi >= G->n_wish) continue;
                if (cfg->share_cap > 0 && rank > cfg->share_cap) continue;
                any = 1;
                const int w = G->wish[wi];
                bstate[nb] = w >= 0 ? G->st.nodes[w].state : G->child_state[-1 - w];
                bgame[nb] = i; bwish[nb] = w; ++nb;
            }
            if (!any) break;
        }
        if (nb == 0) { out->stalls++; break; }            /* cannot happen: a game that is not done and not ready demands its leaf */
        eval(ectx, bstate, nb, pol, val);
        out->launches++; out->rows += nb; out->rows_demanded += demanded_rows;
        for (int i = 0; i < m; ++i) {                      /* children evaluated ahead are stashed per game: drop what an earlier round left */
            fp_game* G = &Gs[i];
            if (G->ahead_leaf >= 0 && G->ahead_leaf != G->leaf) { for (int j = 0; j < G->ahead_n; ++j) free(G->ahead[j].pr); G->ahead_leaf = -1; G->ahead_n = 0; }
        }
        for (int r = 0; r < nb; ++r) {
            fp_game* G = &Gs[bgame[r]];
            if (bwish[r] >= 0) {
                if (!G->ev[bwish[r]].have) ev_fill(g, &G->ev[bwish[r]], &bstate[r], pol + (size_t)r * A, val[r], plays);
                int w;
                if (cfg->preexpand && !G->st.nodes[bwish[r]].drained && !is_terminal(g, &G->st.nodes[bwish[r]].state, &w))
                    fp_expand(g, G, bwish[r], &G->ev[bwish[r]], seed, path_key(G, bwish[r]), NULL, 0.0f, plays, 1);
            }
            else {
                const int j = -1 - bwish[r];
                if (G->ahead_leaf != G->leaf) { G->ahead_leaf = G->leaf; G->ahead_n = 0; memset(G->ahead, 0, sizeof G->ahead); }
                /* children ride in play order; a child that is a finished game was skipped: its slot stays `have = 0` */
                while (G->ahead_n <= j) { memset(&G->ahead[G->ahead_n], 0, sizeof(fp_eval_t)); G->ahead_n++; }
                ev_fill(g, &G->ahead[j], &bstate[r], pol + (size_t)r * A, val[r], plays);
            }
        }
    }
    for (int i = 0; i < m; ++i) {
        fp_game* G = &Gs[i];
        out->rounds_of_game_sum += G->rounds; if (G->rounds > out->rounds_of_game_max) out->rounds_of_game_max = G->rounds;
        for (int k = 0; k < G->ev_cap; ++k) free(G->ev[k].pr);
        for (int j = 0; j < G->ahead_n; ++j) free(G->ahead[j].pr);
        free(G->sh_first); free(G->sh_k);
        free(G->ev); free(G->wish); free(G->child_state); free(G->vvis); free(G->vval); or_store_free(&G->st);
    }
    free(adv); free(order_); free(Gs); free(plays); free(bstate); free(bgame); free(bwish); free(pol); free(val); free(noise);
    return 0;
}

/* states of `n_games` seeded random-walk games at ply `ply` (games that ended earlier are skipped): the live games of move-step `ply`
 * of a self-play batch, near enough (a random-init net plays close to uniformly at random) */
int fp_states_at_ply(uint64_t seed, uint32_t n_games, uint32_t ply, or_bg_state* out, uint32_t* rounds_out) {
    int n = 0;
    or_play* plays = malloc(sizeof(or_play) * OR_MAX_PLAYS);
    for (uint32_t gi = 0; gi < n_games; ++gi) {
        or_bg_state s; or_bg_new(&s);
        uint8_t d0, d1; or_dice(seed, gi, 0, OR_TAG_INIT_ROLL, 0, &d0, &d1);
        s.roll[0] = d0; s.roll[1] = d1;
        int alive = 1;
        for (uint32_t p = 0; p < ply; ++p) {
            int w;
            if (or_bg_check_winner(&s, &w)) { alive = 0; break; }
            int k = or_bg_valid_moves(&s, plays, OR_MAX_PLAYS);
            or_dice(seed, gi, p, OR_TAG_MOVE_ROLL, 0, &d0, &d1);
            if (k == 0) { or_bg_skip_turn(&s, d0, d1); continue; }
            double u = or_uniform01(seed, gi, p, OR_TAG_SAMPLE, 0);
            int j = (int)(u * k); if (j >= k) j = k - 1;
            or_bg_apply_move(&s, &plays[j], d0, d1);
        }
        int w;
        if (alive && !or_bg_check_winner(&s, &w)) { out[n] = s; if (rounds_out) rounds_out[n] = ply; ++n; }
    }
    free(plays);
    return n;
}
