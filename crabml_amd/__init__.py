"""crabml_amd -- MI355X-native likelihood core for crabML's codon models.

The product is the HIP library ``libcrabml_amd.so`` (C ABI: include/crabml_amd.h).
This package is its Python host side:

    crabml_rust        drop-in for the PyO3 module ``crabml.crabml_rust`` (seven functions)
    likelihood_rust    mirror of crabML's ``RustLikelihoodCalculator`` on a persistent engine
    engine             handle API (tree program + compressed alignment resident in HBM)
    models, io         input producers: codon Q matrices, site-class builders, PHYLIP/FASTA/Newick
    synth              seeded synthetic trees / alignments for benchmarks
    dist               pattern sharding across GPUs (one process per GPU, RCCL all-reduce of lnL)
"""
__version__ = "0.1.0"
