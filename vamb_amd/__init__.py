"""vamb_amd -- MI355X-native hot path for Vamb: VAE train -> encode -> medoid clustering.

Public surface mirrors the reference modules it replaces:
    vamb_amd.encode.make_dataloader / set_batchsize / VAE      (vamb/encode.py)
    vamb_amd.cluster.ClusterGenerator / Cluster                 (vamb/cluster.py)
    vamb_amd.dropin.install()                                   binds them onto an imported `vamb`
"""
__version__ = "0.1.0"
