/* oracle/hb_driver.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A minimal driver of OUR OWN that links against the reference's poa-graph
 * objects (built by oracle/Makefile into oracle/_ref/obj, never copied) so the
 * heaviest-bundle consensus can be observed.  The reference compiles
 * generate_lpo_bundles() but its main() never calls it
 * (src/poa-graph/main.c:345-347), so the stock `poa` binary cannot show it.
 *
 * Per (reference, corrected, uncorrected) triple it does what the reference
 * main() does (main.c:265-274: initialize_seqs_as_lpo x3,
 * buildup_progressive_lpo(3, ..., matrix_scoring_function, global=1,
 * preserve_order=1)), then generate_lpo_bundles(lpo, 0.9) (0.9 = main.c:30
 * default bundling_threshold), then write_lpo_bundle_as_fasta(ALL_BUNDLES).
 *
 * usage: poa_hb MATRIX REF.fa COR.fa UNC.fa OUT.fa
 */
#include <stdio.h>
#include <stdlib.h>
#include "lpo.h"
#include "align_score.h"

int main(int argc, char **argv)
{
  ResidueScoreMatrix_T m;
  LPOSequence_T *ref = NULL, *cor = NULL, *unc = NULL, *in[3], *out;
  FILE *f, *o;
  int n, n2, n3, i;
  char *comment = NULL;

  if (argc < 6) { fprintf(stderr, "usage: %s MATRIX REF COR UNC OUT\n", argv[0]); return 2; }
  if (read_score_matrix(argv[1], &m) <= 0) return 1;
  if (!(f = fopen(argv[2], "r"))) return 1;
  n = read_fasta(f, &ref, switch_case_to_lower, &comment); fclose(f);
  if (!(f = fopen(argv[3], "r"))) return 1;
  n2 = read_fasta(f, &cor, switch_case_to_lower, &comment); fclose(f);
  if (!(f = fopen(argv[4], "r"))) return 1;
  n3 = read_fasta(f, &unc, switch_case_to_lower, &comment); fclose(f);
  if (n2 < n || n3 < n) return 1;
  if (!(o = fopen(argv[5], "w"))) return 1;
  for (i = 0; i < n; i++) {
    in[0] = &ref[i]; in[1] = &cor[i]; in[2] = &unc[i];
    initialize_seqs_as_lpo(1, in[0], &m);
    initialize_seqs_as_lpo(1, in[1], &m);
    initialize_seqs_as_lpo(1, in[2], &m);
    out = buildup_progressive_lpo(3, in, &m, 0, 0, NULL, matrix_scoring_function, 1, 1);
    generate_lpo_bundles(out, 0.9f);
    write_lpo_bundle_as_fasta(o, out, m.nsymbol, m.symbol, ALL_BUNDLES);
  }
  fclose(o);
  return 0;
}
