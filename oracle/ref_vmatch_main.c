/*
  TEST INFRASTRUCTURE (oracle/): driver around the REFERENCE matcher.

  This main is ours; everything it calls is compiled straight from
  /root/reference/src by oracle/Makefile.ref (outputs only in oracle/_ref/).
  It stands in for src/Vmatch/vmatch.mn.c:35-102 (whose -version macro needs
  the generated include/vmrelease.h); matching, option parsing, post
  processing and printing are the reference's own callvmatch
  (src/Vmatch/vmatch.c:43) and wrapvmatch.

  Usage: vmatch_ref <vmatch options>       e.g. -complete -d -q Q IDX
  Env:   VMREF_SWALLOW=1  matches go to a no-op sink and a line
                          "# TIME <seconds>" is printed (engine timing without
                          output formatting, like the reference's
                          VMATCHSHOWTIMESPACE, vmatch.mn.c:46-55,91-97).
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "types.h"
#include "errordef.h"
#include "virtualdef.h"
#include "multidef.h"
#include "match.h"
#include "select.h"

void makeemptyvirtualtree(Virtualtree *virtualtree);

Sint callvmatch(Argctype argc,
                const char **argv,
                void *processinfo,
                void initinfo(void *, void *),
                const char *functionname,
                Showmatchfuntype showmatchfun,
                Showverbose showverbose,
                FILE *outfp,
                SelectBundle *precompiledselectbundle,
                Virtualtree *virtualtree,
                Virtualtree *queryvirtualtree,
                Virtualtree *sixframeofqueryvirtualtree,
                Virtualtree *dnavirtualtree);

static void showonstdout(char *s)
{
  printf("# %s\n", s);
}

static Sint swallowmatch(void *showmatchinfo, Multiseq *virtualmultiseq,
                         Multiseq *multiseq, StoreMatch *storematch)
{
  (void) showmatchinfo;
  (void) virtualmultiseq;
  (void) multiseq;
  (void) storematch;
  return 0;
}

static double nowseconds(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

int main(int argc, const char *argv[])
{
  Virtualtree virtualtree, queryvirtualtree, sixframeofqueryvirtualtree,
              dnavirtualtree;
  Sint retcode;
  const char *env = getenv("VMREF_SWALLOW");
  int swallow = (env != NULL && strcmp(env, "1") == 0);
  double t0 = nowseconds();

  makeemptyvirtualtree(&virtualtree);
  makeemptyvirtualtree(&queryvirtualtree);
  makeemptyvirtualtree(&sixframeofqueryvirtualtree);
  makeemptyvirtualtree(&dnavirtualtree);
  retcode = callvmatch(argc, argv, NULL, NULL,
                       swallow ? "swallowmatch" : "NULL",
                       swallow ? swallowmatch : NULL,
                       showonstdout, stdout, NULL,
                       &virtualtree, &queryvirtualtree,
                       &sixframeofqueryvirtualtree, &dnavirtualtree);
  if (retcode < 0)
  {
    fprintf(stderr, "%s: %s\n", argv[0], messagespace());
    return EXIT_FAILURE;
  }
  if (retcode == 0)
  {
    if (wrapvmatch(&virtualtree, &queryvirtualtree,
                   &sixframeofqueryvirtualtree, &dnavirtualtree) != 0)
    {
      fprintf(stderr, "%s: %s\n", argv[0], messagespace());
      return EXIT_FAILURE;
    }
  }
  if (swallow)
  {
    printf("# TIME %.6f\n", nowseconds() - t0);
  }
  return EXIT_SUCCESS;
}
