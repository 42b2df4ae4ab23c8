"""TEST-ONLY scan backend built on the oracle (oracle/cluster_scan.c).

Injected into ``vamb_amd.cluster.ClusterGenerator`` through its private ``_backend_factory`` hook so
the HOST logic (speculative wander, lazy packing, physical/logical indices) can be checked against
the reference's golden cluster streams on a machine without a GPU.  Never used by the product."""
import numpy as np

import cluster_oracle as co
from vamb_amd.cluster import ScanStats


class OracleScanBackend:
    def __init__(self, matrix, lengths_f32, normalized, normalized_out):
        m = np.ascontiguousarray(matrix).copy()
        if not normalized:
            co.normalize(m)
        if normalized_out is not None:
            normalized_out[...] = m
        self.m = m
        self.lengths = np.ascontiguousarray(lengths_f32).copy()
        self.kept = np.ones(len(m), np.uint8)
        self.n_rows = len(m)
        self.L = m.shape[1]
        self.scan_passes = 0
        self.scan_medoids = 0
        self.max_batch = 0

    def scan(self, medoids):
        self.scan_passes += 1
        self.scan_medoids += len(medoids)
        self.max_batch = max(self.max_batch, len(medoids))
        out = []
        for med in medoids:
            r = co.scan(self.m, self.lengths, self.kept, med, want_dist=False)
            out.append(ScanStats(r["density_fx"], r["n_within"], r["n_lt"], r["hist_fx"]))
        return out

    # ---- pieces the row-sharded backend (vamb_amd.parallel.ShardedScanBackend) needs -------------
    def get_rows(self, rows):
        return self.m[np.asarray(rows, dtype=np.int64)].copy()

    def scan_raw(self, rows, queries=None):
        self.scan_passes += 1
        out = np.zeros((len(rows), 63), np.int64)
        for j, med in enumerate(rows):
            q = self.m[med] if queries is None else queries[j]
            r = co.scan_query(self.m, self.lengths, self.kept, int(med), q)
            out[j, 0] = r["density_fx"]
            out[j, 1:61] = r["hist_fx"]
            out[j, 61] = r["n_within"]
            out[j, 62] = r["n_lt"]
        return out

    def select_query(self, row, query, threshold, remove):
        q = self.m[row] if query is None else query
        rows = co.select_query(self.m, self.kept, int(row), q, threshold)
        if remove:
            self.kept[rows] = 0
        return rows

    def select(self, medoid, threshold, remove):
        rows = co.select(self.m, self.kept, medoid, threshold)
        if remove:
            self.kept[rows] = 0
        return rows

    def remove(self, rows):
        self.kept[np.asarray(rows)] = 0

    def pack(self):
        keep = self.kept.astype(bool)
        self.m = np.ascontiguousarray(self.m[keep])
        self.lengths = np.ascontiguousarray(self.lengths[keep])
        self.kept = np.ones(len(self.m), np.uint8)
        self.n_rows = len(self.m)
        return self.n_rows

    def matrix(self):
        return self.m.copy()
