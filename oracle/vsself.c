/*
  TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see vsoracle.h).

  CPU restatement of the reference's supermaximal repeats,
  vmatch -supermax -l L IDX:
    findsupermax / selectsupermaxialrepeats / verifysupermaximality
                                         Vmengine/fsuper.c:60-165
    the depth first traversal            include/vdfstrav.c:247-420
    processexactselfmatch (ACCEPTMATCH)  Vmengine/fself.c:21-38,95-125

  The traversal completes a node [left, right] of depth d with "alwaysontop"
  exactly when all its children are leaves: lcp[left] < d, lcp[left+1 ..
  right] == d, lcp[right+1] < d (vdfstrav.c:330-397 with the edge macros of
  fsuper.c:11-19).  Such a node is reported if d >= L and the characters to
  the left of its suffixes are pairwise different, special symbols not
  counted, the suffix at text position 0 counted once (fsuper.c:60-103); then
  every pair of its suffixes is a match (fsuper.c:105-127), smaller start
  first (ACCEPTMATCH).  Nodes come in suffix array order.
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vsoracle.h"

void orc_push_match(orc_matches *out, uint64_t length, uint64_t dbstart,
                    uint64_t queryseq, uint64_t querystart);

static uint64_t sufat(const orc_index *ix, uint64_t i)
{
  return ix->isize == 4 ? ((const uint32_t *) ix->suf)[i]
                        : ((const uint64_t *) ix->suf)[i];
}

/* lcp value with the exceptions >= 255 from llv (virtualdef.h:121-136) */
static uint64_t lcpat(const orc_index *ix, uint64_t i)
{
  uint64_t lo = 0, hi = ix->nllv;

  if (ix->lcp[i] < 255)
  {
    return ix->lcp[i];
  }
  while (lo < hi)
  {
    const uint64_t mid = lo + (hi - lo) / 2;
    const uint64_t key = ix->isize == 4
                             ? ((const uint32_t *) ix->llv)[2 * mid]
                             : ((const uint64_t *) ix->llv)[2 * mid];
    if (key < i)
    {
      lo = mid + 1;
    } else
    {
      hi = mid;
    }
  }
  return ix->isize == 4 ? ((const uint32_t *) ix->llv)[2 * lo + 1]
                        : ((const uint64_t *) ix->llv)[2 * lo + 1];
}

int orc_findsupermax(const orc_index *ix, uint64_t searchlength,
                     orc_matches *out, char *err)
{
  uint64_t c;

  if (ix->bwt == NULL)
  {
    snprintf(err, 256, "table bwt is not loaded");
    return -1;
  }
  if (ix->hasqueries)
  {
    /* fself.c:193-198 */
    snprintf(err, 256, "supermaximal repeat search does not allow query "
             "files in index");
    return -1;
  }
  if (ix->n < 2)
  {
    snprintf(err, 256, "repeat search requires a sequence of length >= 2");
    return -1;
  }
  /* leaves 0 .. n; lcp[i] belongs to leaves i-1 and i */
  c = 0;
  while (c + 1 <= ix->n)
  {
    const uint64_t d = lcpat(ix, c + 1);
    uint64_t r, q, s, t;
    int marktab[256], marksep = 0, ok = 1;

    if (d < searchlength || d == 0 || (c > 0 && lcpat(ix, c) >= d))
    {
      c++;
      continue;
    }
    r = c + 1;
    while (r + 1 <= ix->n && lcpat(ix, r + 1) == d)
    {
      r++;
    }
    if (r + 1 <= ix->n && lcpat(ix, r + 1) > d)
    {
      c = r; /* a deeper node follows: not all children are leaves */
      continue;
    }
    memset(marktab, 0, sizeof marktab);
    for (q = c; q <= r && ok; q++)
    {
      if (sufat(ix, q) == 0) /* q == longest */
      {
        if (marksep)
        {
          ok = 0;
        }
        marksep = 1;
      } else
      {
        const uint8_t cc = ix->bwt[q];
        if (!ORC_ISSPECIAL(cc))
        {
          if (marktab[cc])
          {
            ok = 0;
          }
          marktab[cc] = 1;
        }
      }
    }
    if (ok)
    {
      for (s = c; s < r; s++)
      {
        for (t = s + 1; t <= r; t++)
        {
          const uint64_t i = sufat(ix, s), j = sufat(ix, t);
          orc_push_match(out, d, i < j ? i : j, i < j ? j : i, 0);
        }
      }
    }
    c = r;
  }
  return 0;
}

/*
  Maximal repeats, vmatch -l L IDX:
    vmatmaxoutgeneric (VMATMAXOUT)   Vmengine/vmatfind.c:487-541
    processleafedge                  Vmengine/vmatfind.c:330-400
    processbranch                    Vmengine/vmatfind.c:413-473
    cartproduct1/2, addtoposlist     Vmengine/vmatfind.c:170-291
    the traversal                    include/vdfstrav.c:247-420
  A node of the lcp-interval tree of depth >= L keeps the start positions of
  the suffixes below it in one list per left character (regular symbols) and
  one list for the others (special symbols, start of the text).  Whenever a
  child -- a leaf or a finished subtree -- is attached to its father, every
  pair (position already at the father, position of the child) with
  different or non-regular left characters is a maximal repeat of length
  depth(father), in the order of the nested loops below.  This is a direct
  simulation with explicit lists (the reference keeps them in shared stacks,
  which is the same thing).
*/

typedef struct
{
  uint64_t *v;
  uint64_t n, cap;
} Vec;

static void vecpush(Vec *a, uint64_t x)
{
  if (a->n == a->cap)
  {
    a->cap = a->cap ? 2 * a->cap : 8;
    a->v = (uint64_t *) realloc(a->v, a->cap * sizeof(uint64_t));
    if (a->v == NULL)
    {
      fprintf(stderr, "oracle: out of memory\n");
      exit(EXIT_FAILURE);
    }
  }
  a->v[a->n++] = x;
}

static void vecappend(Vec *a, const Vec *b)
{
  uint64_t i;
  for (i = 0; i < b->n; i++)
  {
    vecpush(a, b->v[i]);
  }
}

typedef struct
{
  uint64_t depth;
  int lastisleafedge;
  Vec *cls; /* numofchars lists */
  Vec uniq;
} Rnode;

typedef struct
{
  const orc_index *ix;
  uint64_t searchlength, depth;
  uint32_t nc;
  orc_matches *out;
} Rstate;

static void routput(Rstate *st, uint64_t i, uint64_t j)
{
  /* processexactselfmatch: ACCEPTMATCH, fself.c:21-38 */
  const uint64_t start1 = i < j ? i : j, start2 = i < j ? j : i;

  if (st->ix->hasqueries && (start1 >= st->ix->querysepposition ||
                             start2 <= st->ix->querysepposition))
  {
    return; /* only database against query */
  }
  orc_push_match(st->out, st->depth, start1, start2, 0);
}

static void rleafedge(Rstate *st, int firstsucc, Rnode *father,
                      uint32_t leftchar, uint64_t leaf)
{
  uint32_t base;
  uint64_t k;

  if (father->depth < st->searchlength)
  {
    return;
  }
  st->depth = father->depth;
  if (firstsucc)
  {
    for (base = 0; base < st->nc; base++)
    {
      father->cls[base].n = 0;
    }
    father->uniq.n = 0;
  } else
  {
    for (base = 0; base < st->nc; base++)
    {
      if (base != leftchar)
      {
        for (k = 0; k < father->cls[base].n; k++)
        {
          routput(st, leaf, father->cls[base].v[k]);
        }
      }
    }
    for (k = 0; k < father->uniq.n; k++)
    {
      routput(st, leaf, father->uniq.v[k]);
    }
  }
  if (leftchar >= st->nc)
  {
    vecpush(&father->uniq, leaf);
  } else
  {
    vecpush(&father->cls[leftchar], leaf);
  }
}

static void rbranch(Rstate *st, int firstsucc, Rnode *father, Rnode *son)
{
  uint32_t chf, chs;
  uint64_t a, b;

  if (father->depth < st->searchlength || firstsucc)
  {
    return; /* first child: the father took over the son's slot and lists */
  }
  st->depth = father->depth;
  for (chf = 0; chf < st->nc; chf++)
  {
    for (chs = 0; chs < st->nc; chs++)
    {
      if (chs != chf)
      {
        for (a = 0; a < father->cls[chf].n; a++)
        {
          for (b = 0; b < son->cls[chs].n; b++)
          {
            routput(st, father->cls[chf].v[a], son->cls[chs].v[b]);
          }
        }
      }
    }
    for (b = 0; b < son->uniq.n; b++)
    {
      for (a = 0; a < father->cls[chf].n; a++)
      {
        routput(st, son->uniq.v[b], father->cls[chf].v[a]);
      }
    }
  }
  for (a = 0; a < father->uniq.n; a++)
  {
    for (chs = 0; chs < st->nc; chs++)
    {
      for (b = 0; b < son->cls[chs].n; b++)
      {
        routput(st, father->uniq.v[a], son->cls[chs].v[b]);
      }
    }
    for (b = 0; b < son->uniq.n; b++)
    {
      routput(st, father->uniq.v[a], son->uniq.v[b]);
    }
  }
  for (chs = 0; chs < st->nc; chs++)
  {
    vecappend(&father->cls[chs], &son->cls[chs]);
  }
  vecappend(&father->uniq, &son->uniq);
}

int orc_findmaximalrepeats(const orc_index *ix, uint64_t searchlength,
                           orc_matches *out, char *err)
{
  Rnode *stack = NULL;
  uint64_t allocated = 0, top = 0, c, k; /* top = number of nodes */
  Rstate st;
  int firstrootedge = 1;

  if (ix->bwt == NULL)
  {
    snprintf(err, 256, "table bwt is not loaded");
    return -1;
  }
  if (ix->n < 2)
  {
    snprintf(err, 256, "repeat search requires a sequence of length >= 2");
    return -1;
  }
  st.ix = ix;
  st.searchlength = searchlength;
  st.depth = 0;
  st.nc = ix->numofchars;
  st.out = out;
#define NEEDSLOT(I)                                                           \
  while ((I) >= allocated)                                                    \
  {                                                                           \
    const uint64_t na = allocated ? 2 * allocated : 64;                       \
    stack = (Rnode *) realloc(stack, na * sizeof(Rnode));                     \
    for (k = allocated; k < na; k++)                                          \
    {                                                                         \
      memset(stack + k, 0, sizeof(Rnode));                                    \
      stack[k].cls = (Vec *) calloc(st.nc, sizeof(Vec));                      \
    }                                                                         \
    allocated = na;                                                           \
  }
  NEEDSLOT(1);
  stack[0].depth = 0;
  stack[0].lastisleafedge = 1;
  top = 1;
  for (c = 0; c + 1 <= ix->n; c++) /* leaves 0 .. n-1; lcp[c+1] follows c */
  {
    const uint64_t currentlcp = lcpat(ix, c + 1), leaf = sufat(ix, c);
    const uint32_t leftchar = (leaf == 0) ? st.nc + 1 : ix->bwt[c];

    while (currentlcp < stack[top - 1].depth)
    {
      if (stack[top - 1].lastisleafedge)
      {
        rleafedge(&st, 0, stack + top - 1, leftchar, leaf);
      } else
      {
        rbranch(&st, 0, stack + top - 1, stack + top);
      }
      top--;
    }
    if (currentlcp == stack[top - 1].depth)
    {
      int firstedge = 0;
      if (firstrootedge && stack[top - 1].depth == 0)
      {
        firstedge = 1;
        firstrootedge = 0;
      }
      if (stack[top - 1].lastisleafedge)
      {
        rleafedge(&st, firstedge, stack + top - 1, leftchar, leaf);
      } else
      {
        rbranch(&st, firstedge, stack + top - 1, stack + top);
        stack[top - 1].lastisleafedge = 1;
      }
    } else
    {
      NEEDSLOT(top + 1);
      stack[top].depth = currentlcp;
      stack[top].lastisleafedge = 1;
      top++;
      if (stack[top - 2].lastisleafedge)
      {
        rleafedge(&st, 1, stack + top - 1, leftchar, leaf);
        stack[top - 2].lastisleafedge = 0;
      } /* else: the slot still holds the lists of the son just finished */
    }
  }
  /* the last leaf n hangs below the root (lcp[n] = 0): nothing to report */
  for (k = 0; k < allocated; k++)
  {
    uint32_t b;
    for (b = 0; b < st.nc; b++)
    {
      free(stack[k].cls[b].v);
    }
    free(stack[k].cls);
    free(stack[k].uniq.v);
  }
  free(stack);
#undef NEEDSLOT
  return 0;
}

/*
  Right branching tandem repeats, vmatch -tandem -l L IDX:
    findtandems / processcompletenode / processsmallinterval /
    tandemleftright                      Vmengine/ftandem.c:24-304
    the depth first traversal            include/vdfstrav.c:247-420
    findmaxprefixlen                     kurtz/findmaxpref.gen:1-96

  Every lcp-interval [left, right] of depth d >= L is handed over when the
  traversal completes it (children first, left to right).  Two suffixes: a
  tandem if one starts d symbols behind the other (CHECKPAIR :60-71).  More:
  the suffixes below the node that begin with the node's string twice are
  looked up with one call of findmaxprefixlen (query = the text d symbols in
  front of suf[left], of which the first d count as matched, :235-241) and
  enumerated from the witness, first to the left, then to the right
  (tandemleftright :105-186).  A repeat of length d at v is reported unless
  the symbol behind it equals the one at v and both are regular -- then it
  shifts right and is reported there (SHOWTANDEM :44-58, PROCESSSUFFIX
  :73-88).  length = d, dbstart = v, queryseq = v + d, querystart = 0.
*/
typedef struct
{
  uint64_t depth, left;
} Tnode;

static void tandemout(const orc_index *ix, uint64_t d, uint64_t v, uint64_t w,
                      orc_matches *out)
{
  /* SHOWTANDEM(v, w) with w = v + d; PROCESSSUFFIX with w = v + d too */
  if (w + d == ix->n)
  {
    orc_push_match(out, d, v, v + d, 0);
  } else
  {
    const uint8_t c1 = ix->tis[v], c2 = ix->tis[w + d];
    if (c1 != c2 || ORC_ISSPECIAL(c1) || ORC_ISSPECIAL(c2))
    {
      orc_push_match(out, d, v, v + d, 0);
    }
  }
}

/* COMPARE of kurtz/maxpref.c:30-65 for a query that is the text at qpos */
static int tcompare(const orc_index *ix, uint64_t sufstart, int64_t qpos,
                    uint64_t querylen, uint64_t *lcplen)
{
  uint64_t l = *lcplen;
  int ret;

  for (;; l++)
  {
    uint8_t q;
    if (l >= querylen)
    {
      ret = 0;
      break;
    }
    if (sufstart + l >= ix->n)
    {
      ret = -1;
      break;
    }
    q = ix->tis[qpos + (int64_t) l];
    ret = (int) q - (int) ix->tis[sufstart + l];
    if (ret == 0)
    {
      if (ORC_ISSPECIAL(q))
      {
        ret = -1;
        break;
      }
    } else
    {
      break;
    }
  }
  *lcplen = l;
  return ret;
}

static void tfindmaxprefixlen(const orc_index *ix, uint64_t vleft,
                              uint64_t vright, uint64_t offset, int64_t qpos,
                              uint64_t querylen, uint64_t *maxlcp,
                              uint64_t *witness)
{
  uint64_t left, right, mid, lcplen, lpref, rpref;
  int ret;

  lcplen = offset;
  ret = tcompare(ix, sufat(ix, vleft), qpos, querylen, &lcplen);
  *maxlcp = lcplen;
  *witness = vleft;
  if (ret <= 0)
  {
    return;
  }
  lpref = lcplen;
  lcplen = offset;
  ret = tcompare(ix, sufat(ix, vright), qpos, querylen, &lcplen);
  rpref = lcplen;
  if (lpref < rpref)
  {
    *maxlcp = rpref;
    *witness = vright;
    lcplen = lpref;
  } else
  {
    *maxlcp = lpref;
    *witness = vleft;
  }
  if (ret >= 0 || *maxlcp >= querylen)
  {
    return;
  }
  left = vleft;
  right = vright;
  while (right > left + 1)
  {
    mid = (left + right) / 2;
    ret = tcompare(ix, sufat(ix, mid), qpos, querylen, &lcplen);
    if (*maxlcp < lcplen)
    {
      *maxlcp = lcplen;
      *witness = mid;
    }
    if (ret < 0)
    {
      rpref = lcplen;
      if (lpref < rpref)
      {
        lcplen = lpref;
      }
      right = mid;
    } else if (ret > 0)
    {
      lpref = lcplen;
      if (rpref < lpref)
      {
        lcplen = rpref;
      }
      left = mid;
    } else
    {
      break;
    }
  }
}

static void tandemnode(const orc_index *ix, uint64_t d, uint64_t left,
                       uint64_t right, orc_matches *out)
{
  if (right - left + 1 <= 2)
  {
    const uint64_t s0 = sufat(ix, left), s1 = sufat(ix, left + 1);
    if (s0 + d == s1)
    {
      tandemout(ix, d, s0, s1, out);
    } else if (s1 + d == s0)
    {
      tandemout(ix, d, s1, s0, out);
    }
  } else
  {
    uint64_t maxlcp, witness, ind;
    tfindmaxprefixlen(ix, left, right, d,
                      (int64_t) sufat(ix, left) - (int64_t) d, 2 * d, &maxlcp,
                      &witness);
    if (maxlcp != 2 * d)
    {
      return;
    }
    for (ind = witness;; ind--)
    {
      const uint64_t s = sufat(ix, ind);
      tandemout(ix, d, s, s + d, out);
      if (ind == 0 || lcpat(ix, ind) < 2 * d)
      {
        break;
      }
    }
    for (ind = witness + 1;; ind++)
    {
      uint64_t s;
      if (ind > ix->n || lcpat(ix, ind) < 2 * d)
      {
        break;
      }
      s = sufat(ix, ind);
      tandemout(ix, d, s, s + d, out);
    }
  }
}

int orc_findtandems(const orc_index *ix, uint64_t searchlength,
                    orc_matches *out, char *err)
{
  Tnode *stack = NULL;
  uint64_t allocated = 0, top = 0, c;

  if (ix->hasqueries)
  {
    snprintf(err, 256,
             "tandem repeat search does not allow query files in index");
    return -1;
  }
  if (searchlength == 0)
  {
    snprintf(err, 256, "tandem repeat search needs a length >= 1");
    return -1;
  }
  allocated = 64;
  stack = (Tnode *) malloc(allocated * sizeof(Tnode));
  stack[0].depth = 0;
  stack[0].left = 0;
  top = 1;
  for (c = 0; c + 1 <= ix->n; c++) /* lcp[c+1] decides what happens behind c */
  {
    const uint64_t currentlcp = lcpat(ix, c + 1);
    uint64_t newleft = c; /* a leaf edge becomes the first child */

    while (currentlcp < stack[top - 1].depth)
    {
      if (stack[top - 1].depth >= searchlength)
      {
        tandemnode(ix, stack[top - 1].depth, stack[top - 1].left, c, out);
      }
      newleft = stack[top - 1].left; /* the finished subtree is the child */
      top--;
    }
    if (currentlcp > stack[top - 1].depth)
    {
      if (top == allocated)
      {
        allocated *= 2;
        stack = (Tnode *) realloc(stack, allocated * sizeof(Tnode));
      }
      stack[top].depth = currentlcp;
      stack[top].left = newleft;
      top++;
    }
  }
  /* what is left has depth 0 (lcp[n] = 0): below every L */
  free(stack);
  return 0;
}
