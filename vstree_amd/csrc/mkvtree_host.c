/*
  mkvtree on the GPU, host side (plain C): FASTA files in, the index files of
  `mkvtree -db F.. [-q Q..] -dna -pl [n] -allout` out, byte for byte.

  What the reference does on one CPU core in Mkvtree/mkvinput.c (input),
  Mkvtree/ppsort.c + Mkvtree/bese.c (sorting, lcp) and
  Mkvtree/mkvprocess.c:99-816 (table writers) is split here into
    * this file: reading multiple FASTA (kurtz-basic/multiseq-adv.c:1033-1250
      semantics: '>' at the start of a line opens a description, white space is
      skipped, every other character goes through the symbol map, an unknown
      one is the reference's "Illegal character" error; sequences are joined
      by SEPARATOR), the DNA symbol map of mkvtree -dna, bookkeeping for the
      .prj file, and writing the files;
    * index_build.hip: suf, lcp/llv, bck, bwt, sti1 on the GPU.

  Files written (raw host-endian arrays, no headers; integersize = 64 like
  the reference's LP64 build, or 32 on request):
    .prj .al1 .tis .ois .des .sds .ssp .suf .lcp .llv .bck .bwt .sti1 [.skp]
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <ctype.h>
#include <errno.h>
#include "vstree_amd.h"

char *vsa_errbuf(void);
#define ERRSIZE 1024

typedef struct
{
  uint8_t *tis, *ois;      /* mapped / original symbols, n each            */
  uint64_t n, cap, oiscap;
  char *des;               /* descriptions, each ending in '\n'            */
  uint64_t deslen, descap;
  uint64_t *sds, *ssp;     /* description starts [numseq+1], separators    */
  uint64_t numseq, sdscap, sspcap;
  uint64_t specialcharacters, specialranges;
} Text;

static int grow(void **p, uint64_t *cap, uint64_t need, size_t elem)
{
  if (need > *cap)
  {
    uint64_t nc = *cap ? *cap : 4096;
    void *q;
    while (nc < need)
    {
      nc *= 2;
    }
    q = realloc(*p, (size_t) nc * elem);
    if (q == NULL)
    {
      snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
      return -1;
    }
    *p = q;
    *cap = nc;
  }
  return 0;
}

/* symbol map of mkvtree -dna: what it writes to IDX.al1 */
static const char *dna_al1 = "aA\ncC\ngG\ntTuU\nnsywrkvbdhmNSYWRKVBDHM\n";

static void dnasymbolmap(uint8_t map[256])
{
  const char *classes[4] = {"aA", "cC", "gG", "tTuU"};
  const char *wild = "nsywrkvbdhmNSYWRKVBDHM";
  int c;
  const char *p;

  memset(map, VSA_UNDEFBWT, 256); /* 253 = undefined symbol */
  for (c = 0; c < 4; c++)
  {
    for (p = classes[c]; *p != '\0'; p++)
    {
      map[(uint8_t) *p] = (uint8_t) c;
    }
  }
  for (p = wild; *p != '\0'; p++)
  {
    map[(uint8_t) *p] = (uint8_t) VSA_WILDCARD;
  }
}

static int pushsym(Text *t, uint8_t mapped, uint8_t orig)
{
  if (grow((void **) &t->tis, &t->cap, t->n + 1, 1) != 0 ||
      grow((void **) &t->ois, &t->oiscap, t->n + 1, 1) != 0)
  {
    return -1;
  }
  t->tis[t->n] = mapped;
  t->ois[t->n] = orig;
  t->n++;
  return 0;
}

/* one FASTA file appended to the text; *filelength = bytes in the file */
static int readfasta(Text *t, const char *path, const uint8_t map[256],
                     uint64_t *filelength)
{
  FILE *fp = fopen(path, "rb");
  int ch, prev = '\n', indesc = 0, seenseq = 0;
  uint64_t linenum = 1;

  if (fp == NULL)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "cannot open file \"%s\": %s", path,
             strerror(errno));
    return -1;
  }
  *filelength = 0;
  while ((ch = getc(fp)) != EOF)
  {
    (*filelength)++;
    if (indesc)
    {
      if (grow((void **) &t->des, &t->descap, t->deslen + 1, 1) != 0)
      {
        fclose(fp);
        return -1;
      }
      t->des[t->deslen++] = (char) ch;
      if (ch == '\n')
      {
        linenum++;
        indesc = 0;
      }
    } else if (ch == '>')
    {
      if (prev != '\n')
      {
        snprintf(vsa_errbuf(), ERRSIZE,
                 "Illegal character '%c' in file \"%s\" line %lu", ch, path,
                 (unsigned long) linenum);
        fclose(fp);
        return -1;
      }
      if (grow((void **) &t->sds, &t->sdscap, t->numseq + 2, 8) != 0 ||
          grow((void **) &t->ssp, &t->sspcap, t->numseq + 2, 8) != 0)
      {
        fclose(fp);
        return -1;
      }
      t->sds[t->numseq] = t->deslen;
      if (t->numseq > 0)
      {
        t->ssp[t->numseq - 1] = t->n;
        if (pushsym(t, (uint8_t) VSA_SEPARATOR, (uint8_t) VSA_SEPARATOR) != 0)
        {
          fclose(fp);
          return -1;
        }
      }
      t->numseq++;
      indesc = 1;
      seenseq = 1;
    } else if (isspace(ch))
    {
      if (ch == '\n')
      {
        linenum++;
      }
    } else
    {
      const uint8_t code = map[(uint8_t) ch];
      if (code == VSA_UNDEFBWT || !seenseq)
      {
        snprintf(vsa_errbuf(), ERRSIZE,
                 "Illegal character '%c' in file \"%s\" line %lu", ch, path,
                 (unsigned long) linenum);
        fclose(fp);
        return -1;
      }
      if (pushsym(t, code, (uint8_t) ch) != 0)
      {
        fclose(fp);
        return -1;
      }
    }
    prev = ch;
  }
  fclose(fp);
  return 0;
}

static int writefile(const char *indexname, const char *suffix,
                     const void *data, size_t bytes)
{
  char path[4096 + 32];
  FILE *fp;

  snprintf(path, sizeof path, "%s.%s", indexname, suffix);
  fp = fopen(path, "wb");
  if (fp == NULL)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "cannot open file \"%s\": %s", path,
             strerror(errno));
    return -1;
  }
  if (bytes > 0 && fwrite(data, 1, bytes, fp) != bytes)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "cannot write %lu bytes to \"%s\"",
             (unsigned long) bytes, path);
    fclose(fp);
    return -1;
  }
  if (fclose(fp) != 0)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "cannot close \"%s\"", path);
    return -1;
  }
  return 0;
}

/* entry i of a table of w-byte integers (w = 4 or 8: the width of the device
   tables, 8 for texts of 2^32 symbols and more) */
static uint64_t geti(const void *v, uint32_t w, uint64_t i)
{
  return w == 4 ? (uint64_t) ((const uint32_t *) v)[i]
                : ((const uint64_t *) v)[i];
}

static void seti(void *v, uint32_t w, uint64_t i, uint64_t x)
{
  if (w == 4)
  {
    ((uint32_t *) v)[i] = (uint32_t) x;
  } else
  {
    ((uint64_t *) v)[i] = x;
  }
}

/* table of w-byte entries written as `bits`-wide integers */
static int writeintegers(const char *indexname, const char *suffix,
                         const void *v, uint32_t w, uint64_t count,
                         uint32_t bits)
{
  int rc;
  uint64_t i;
  void *out;

  if (bits == 8 * w)
  {
    return writefile(indexname, suffix, v, (size_t) count * w);
  }
  out = malloc((size_t) (count ? count : 1) * (bits / 8));
  if (out == NULL)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
    return -1;
  }
  for (i = 0; i < count; i++)
  {
    const uint64_t x = geti(v, w, i);
    if (bits == 32 && (x >> 32) != 0)
    {
      snprintf(vsa_errbuf(), ERRSIZE,
               "integersize 32 cannot hold the entries of %s.%s", indexname,
               suffix);
      free(out);
      return -1;
    }
    seti(out, bits / 8, i, x);
  }
  rc = writefile(indexname, suffix, out, (size_t) count * (bits / 8));
  free(out);
  return rc;
}

/* kurtz/mkskip.c:51-96: skp[i] = last index up to which the lcp values stay
   >= lcp[i] (next smaller value to the right, minus one) */
static int makeskiptable(void *skp, const uint8_t *lcp, const void *llv,
                         uint32_t w, uint64_t n)
{
  uint64_t i, top = 0, exception = 0, cap = 1024;
  uint64_t *depth = (uint64_t *) malloc(cap * 8),
           *step = (uint64_t *) malloc(cap * 8);

  if (depth == NULL || step == NULL)
  {
    free(depth);
    free(step);
    snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
    return -1;
  }
  depth[0] = 0;
  step[0] = 0;
  top = 1;
  for (i = 1; i <= n; i++)
  {
    uint64_t cur = lcp[i];
    if (cur == 255)
    {
      cur = geti(llv, w, 2 * (exception++) + 1);
    }
    while (cur < depth[top - 1])
    {
      seti(skp, w, step[top - 1], i - 1);
      top--;
    }
    if (top == cap)
    {
      cap *= 2;
      depth = (uint64_t *) realloc(depth, cap * 8);
      step = (uint64_t *) realloc(step, cap * 8);
      if (depth == NULL || step == NULL)
      {
        snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
        return -1;
      }
    }
    depth[top] = cur;
    step[top] = i;
    top++;
  }
  while (top > 0)
  {
    seti(skp, w, step[top - 1], n);
    top--;
  }
  free(depth);
  free(step);
  return 0;
}

int vsa_mkvtree(const char *const *dbfiles, uint32_t numofdbfiles,
                const char *const *queryfiles, uint32_t numofqueryfiles,
                const char *indexname, uint32_t prefixlength,
                uint32_t integersize, int withskp, int device)
{
  Text t;
  uint8_t map[256];
  uint64_t *filelen = NULL, *fileend = NULL, numofdbsequences = 0, i;
  uint32_t f, nfiles = numofdbfiles + numofqueryfiles;
  vsa_index *ix = NULL;
  vsa_index_info info;
  uint8_t *lcp = NULL, *bwt = NULL, *sti1 = NULL;
  void *suf = NULL, *llv = NULL, *bck = NULL, *skp = NULL;
  uint32_t w = 4; /* bytes per entry of the device tables */
  int rc = -1;
  FILE *prj;
  char path[4096 + 32];

  if (dbfiles == NULL || numofdbfiles == 0 || indexname == NULL ||
      (numofqueryfiles > 0 && queryfiles == NULL))
  {
    snprintf(vsa_errbuf(), ERRSIZE, "vsa_mkvtree: missing argument");
    return -1;
  }
  if (integersize != 32 && integersize != 64)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "integersize must be 32 or 64");
    return -1;
  }
  memset(&t, 0, sizeof t);
  dnasymbolmap(map);
  filelen = (uint64_t *) calloc(nfiles, 8);
  fileend = (uint64_t *) calloc(nfiles, 8);
  if (filelen == NULL || fileend == NULL)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
    goto done;
  }
  for (f = 0; f < nfiles; f++)
  {
    const char *p = f < numofdbfiles ? dbfiles[f]
                                     : queryfiles[f - numofdbfiles];
    if (readfasta(&t, p, map, &filelen[f]) != 0)
    {
      goto done;
    }
    fileend[f] = t.n; /* = position of the separator to the next file */
    if (f + 1 == numofdbfiles)
    {
      numofdbsequences = t.numseq;
    }
  }
  if (t.numseq == 0)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "no sequences in multiple fasta file");
    goto done;
  }
  if (grow((void **) &t.sds, &t.sdscap, t.numseq + 2, 8) != 0)
  {
    goto done;
  }
  t.sds[t.numseq] = t.deslen;
  for (i = 0; i < t.n; i++)
  {
    if (t.tis[i] >= VSA_WILDCARD)
    {
      t.specialcharacters++;
      if (i == 0 || t.tis[i - 1] < VSA_WILDCARD)
      {
        t.specialranges++;
      }
    }
  }
  /* the GPU part */
  if (vsa_index_build(t.tis, t.n, 4, prefixlength, device, &ix) != 0 ||
      vsa_index_getinfo(ix, &info) != 0)
  {
    goto done;
  }
  w = info.device_integersize / 8;
  suf = malloc((size_t) (t.n + 1) * w);
  lcp = (uint8_t *) malloc((size_t) t.n + 1);
  bwt = (uint8_t *) malloc((size_t) t.n + 1);
  sti1 = (uint8_t *) malloc((size_t) t.n + 1);
  llv = malloc((size_t) (2 * info.largelcpvalues + 1) * w);
  bck = malloc((size_t) (2 * info.numofcodes) * w);
  if (suf == NULL || lcp == NULL || bwt == NULL || sti1 == NULL ||
      llv == NULL || bck == NULL)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
    goto done;
  }
  if (vsa_index_download(ix, NULL, suf, lcp, llv, bck, bwt) != 0 ||
      vsa_index_make_sti1(ix, sti1) != 0)
  {
    goto done;
  }
  /* .prj (Mkvtree/mkvprocess.c:403-504) */
  snprintf(path, sizeof path, "%s.prj", indexname);
  prj = fopen(path, "w");
  if (prj == NULL)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "cannot open file \"%s\": %s", path,
             strerror(errno));
    goto done;
  }
  {
    uint64_t startseq = 0, longest = 0, maxbranchdepth = 0, k,
             prefix = 0, suffix = 0;
    for (f = 0; f < nfiles; f++)
    {
      const char *p = f < numofdbfiles ? dbfiles[f]
                                       : queryfiles[f - numofdbfiles];
      fprintf(prj, "%s=%s %lu %lu\n", f < numofdbfiles ? "dbfile"
                                                       : "queryfile",
              p, (unsigned long) filelen[f],
              (unsigned long) (fileend[f] - startseq));
      startseq = fileend[f] + 1;
    }
    for (k = 0; k <= t.n; k++)
    {
      if (geti(suf, w, k) == 0)
      {
        longest = k; /* determinelongest, mkvprocess.c:857-873 */
        break;
      }
    }
    for (k = 0; k <= t.n; k++)
    {
      if (lcp[k] < 255 && lcp[k] > maxbranchdepth)
      {
        maxbranchdepth = lcp[k];
      }
    }
    for (k = 0; k < info.largelcpvalues; k++)
    {
      if (geti(llv, w, 2 * k + 1) > maxbranchdepth)
      {
        maxbranchdepth = geti(llv, w, 2 * k + 1);
      }
    }
    while (prefix < t.n && t.tis[prefix] >= VSA_WILDCARD)
    {
      prefix++;
    }
    while (suffix < t.n && t.tis[t.n - 1 - suffix] >= VSA_WILDCARD)
    {
      suffix++;
    }
    fprintf(prj, "totallength=%lu\n", (unsigned long) t.n);
    fprintf(prj, "specialcharacters=%lu\n",
            (unsigned long) t.specialcharacters);
    fprintf(prj, "specialranges=%lu\n", (unsigned long) t.specialranges);
    fprintf(prj, "lengthofspecialprefix=%lu\n", (unsigned long) prefix);
    fprintf(prj, "lengthofspecialsuffix=%lu\n", (unsigned long) suffix);
    fprintf(prj, "numofsequences=%lu\n", (unsigned long) t.numseq);
    fprintf(prj, "numofdbsequences=%lu\n",
            (unsigned long) numofdbsequences);
    fprintf(prj, "numofquerysequences=%lu\n",
            (unsigned long) (t.numseq - numofdbsequences));
    fprintf(prj, "longest=%lu\n", (unsigned long) longest);
    fprintf(prj, "prefixlength=%lu\n", (unsigned long) info.prefixlength);
    fprintf(prj, "largelcpvalues=%lu\n",
            (unsigned long) info.largelcpvalues);
    fprintf(prj, "maxbranchdepth=%lu\n", (unsigned long) maxbranchdepth);
    fprintf(prj, "integersize=%u\n", integersize);
    {
      const uint16_t probe = 1;
      fprintf(prj, "littleendian=%c\n",
              *(const uint8_t *) &probe == 1 ? '1' : '0');
    }
  }
  if (fclose(prj) != 0)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "cannot close \"%s\"", path);
    goto done;
  }
  if (writefile(indexname, "al1", dna_al1, strlen(dna_al1)) != 0 ||
      writefile(indexname, "tis", t.tis, (size_t) t.n) != 0 ||
      writefile(indexname, "ois", t.ois, (size_t) t.n) != 0 ||
      writefile(indexname, "des", t.des, (size_t) t.deslen) != 0 ||
      writefile(indexname, "lcp", lcp, (size_t) t.n + 1) != 0 ||
      writefile(indexname, "bwt", bwt, (size_t) t.n + 1) != 0 ||
      writefile(indexname, "sti1", sti1, (size_t) t.n + 1) != 0 ||
      writeintegers(indexname, "suf", suf, w, t.n + 1, integersize) != 0 ||
      writeintegers(indexname, "llv", llv, w, 2 * info.largelcpvalues,
                    integersize) != 0 ||
      writeintegers(indexname, "bck", bck, w, 2 * info.numofcodes,
                    integersize) != 0)
  {
    goto done;
  }
  /* .sds and .ssp hold Uint values, too */
  {
    uint64_t *tmp = (uint64_t *) malloc((size_t) (t.numseq + 2) * 8);
    if (tmp == NULL)
    {
      snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
      goto done;
    }
    for (i = 0; i <= t.numseq; i++)
    {
      tmp[i] = t.sds[i];
    }
    rc = writeintegers(indexname, "sds", tmp, 8, t.numseq + 1, integersize);
    for (i = 0; i + 1 < t.numseq; i++)
    {
      tmp[i] = t.ssp[i];
    }
    if (rc == 0 && t.numseq > 1)
    {
      rc = writeintegers(indexname, "ssp", tmp, 8, t.numseq - 1, integersize);
    }
    free(tmp);
    if (rc != 0)
    {
      rc = -1;
      goto done;
    }
    rc = -1;
  }
  if (withskp)
  {
    skp = malloc((size_t) (t.n + 1) * w);
    if (skp == NULL)
    {
      snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
      goto done;
    }
    if (makeskiptable(skp, lcp, llv, w, t.n) != 0 ||
        writeintegers(indexname, "skp", skp, w, t.n + 1, integersize) != 0)
    {
      goto done;
    }
  }
  rc = 0;
done:
  vsa_index_close(ix);
  free(t.tis);
  free(t.ois);
  free(t.des);
  free(t.sds);
  free(t.ssp);
  free(filelen);
  free(fileend);
  free(suf);
  free(lcp);
  free(bwt);
  free(sti1);
  free(llv);
  free(bck);
  free(skp);
  return rc;
}
