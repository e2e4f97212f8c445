/*
  TEST INFRASTRUCTURE (oracle/): driver around the REFERENCE index builder.

  This main is ours; everything it calls is compiled straight from
  /root/reference/src by oracle/Makefile.ref (outputs only in oracle/_ref/).
  It replaces the reference's own main (src/Mkvtree/mkvfile.c:37-81) only
  because that one pulls in the generated header include/vmrelease.h through
  the -version macro; the work is done by the reference's callmkvtree
  (src/Mkvtree/mkvtree.c:689), exactly as in mkvtree.x.

  Usage: mkvtree_ref <mkvtree options>     e.g. -db g.fna -dna -pl -allout
*/
#include <stdio.h>
#include <stdlib.h>
#include "types.h"
#include "errordef.h"
#include "virtualdef.h"

void makeemptyvirtualtree(Virtualtree *virtualtree);
Sint freevirtualtree(Virtualtree *virtualtree);

static void showonstdout(char *s)
{
  printf("%s\n", s);
}

int main(int argc, const char *argv[])
{
  Virtualtree virtualtree;
  Sint ret;

  makeemptyvirtualtree(&virtualtree);
  ret = callmkvtree(argc, argv, True, &virtualtree, True, showonstdout);
  if (ret == (Sint) 1)
  {
    return EXIT_SUCCESS;
  }
  if (ret < 0)
  {
    fprintf(stderr, "%s: %s\n", argv[0], messagespace());
    return EXIT_FAILURE;
  }
  if (freevirtualtree(&virtualtree) != 0)
  {
    fprintf(stderr, "%s: %s\n", argv[0], messagespace());
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
