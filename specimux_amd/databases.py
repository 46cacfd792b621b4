"""Primer / specimen registries of the drop-in interface (reference: src/specimux/databases.py).

Same public surface (PrimerDatabase :17, Specimens :123, BarcodePrefilter :311,
PassthroughPrefilter :319); the lookups the hot path needs are flattened into index arrays by
specimux_amd/panel.py and answered on the GPU."""
import logging
from typing import Dict, List, Optional, Protocol

from .constants import Primer
from .models import PrimerInfo


class PrimerDatabase:
    """name -> PrimerInfo, pool -> primers by direction, in primers.fasta order."""

    def __init__(self):
        self._primers: Dict[str, PrimerInfo] = {}
        self._pools: Dict[str, set] = {}
        self._pool_primers: Dict[str, Dict[Primer, List[PrimerInfo]]] = {}

    def add_primer(self, primer: PrimerInfo, pools: List[str]) -> None:
        if primer.name in self._primers:
            raise ValueError(f"Duplicate primer name: {primer.name}")
        self._primers[primer.name] = primer
        for pool in pools:
            by_dir = self._pool_primers.setdefault(pool, {Primer.FWD: [], Primer.REV: []})
            self._pools.setdefault(pool, set()).add(primer.name)
            by_dir[primer.direction].append(primer)

    def get_primer(self, name: str) -> Optional[PrimerInfo]:
        return self._primers.get(name)

    def get_pools(self) -> List[str]:
        return list(self._pools)

    def get_pool_primers(self, pool: str, direction: Optional[Primer] = None) -> List[PrimerInfo]:
        by_dir = self._pool_primers.get(pool)
        if by_dir is None:
            return []
        return by_dir[direction] if direction else by_dir[Primer.FWD] + by_dir[Primer.REV]

    def get_primers_in_pool(self, pool: str) -> List[PrimerInfo]:
        return self.get_pool_primers(pool)

    def primer_in_pool(self, primer_name: str, pool: str) -> bool:
        return primer_name in self._pools.get(pool, ())

    def validate_pools(self) -> None:
        for pool, by_dir in self._pool_primers.items():
            for direction, word in ((Primer.FWD, "forward"), (Primer.REV, "reverse")):
                if not by_dir[direction]:
                    raise ValueError(f"Pool {pool} has no {word} primers")

    def get_pool_stats(self) -> Dict:
        pools = {p: {"forward_primers": len(d[Primer.FWD]), "reverse_primers": len(d[Primer.REV]),
                     "total_primers": len(d[Primer.FWD]) + len(d[Primer.REV])}
                 for p, d in self._pool_primers.items()}
        return {"total_primers": len(self._primers), "total_pools": len(self._pools), "pools": pools}

    def drop_pool(self, pool: str) -> None:
        self._pools.pop(pool, None)
        self._pool_primers.pop(pool, None)


class Specimens:
    """Specimen table in file order plus the primer registry keyed by primer SEQUENCE.

    Registration order of `_primers` (forward list then reverse list of each specimens.txt row, first
    time a sequence is seen) drives candidate enumeration order and every tie-break downstream
    (SURVEY Q5); it is preserved exactly."""

    def __init__(self, primer_registry: PrimerDatabase):
        self._specimens = []          # (id, pool, b1, [PrimerInfo], b2, [PrimerInfo])
        self._barcode_length = 0
        self._primers: Dict[str, PrimerInfo] = {}
        self._specimen_ids = set()
        self._primer_pairings: Dict[str, List[PrimerInfo]] = {}
        self._primer_registry = primer_registry
        self._active_pools = set()
        self._pool_of: Dict[str, str] = {}

    def _resolve_primer_name(self, primer_name: str, pool: str, direction: Primer) -> List[PrimerInfo]:
        if primer_name in ("-", "*"):
            found = [p for p in self._primer_registry.get_primers_in_pool(pool) if p.direction == direction]
            if not found:
                raise ValueError(f"No {direction.name} primers found in pool {pool}")
            return found
        primer = self._primer_registry.get_primer(primer_name)
        if primer is None:
            raise ValueError(f"Primer not found: {primer_name}")
        if primer.direction != direction:
            raise ValueError(f"Primer {primer_name} is not a {direction.name} primer")
        if not self._primer_registry.primer_in_pool(primer_name, pool):
            raise ValueError(f"Primer {primer_name} is not in pool {pool}")
        return [primer]

    def add_specimen(self, specimen_id: str, pool: str, b1: str, p1: str, b2: str, p2: str):
        if specimen_id in self._specimen_ids:
            raise ValueError(f"Duplicate specimen id in index file: {specimen_id}")
        self._specimen_ids.add(specimen_id)
        self._active_pools.add(pool)
        self._pool_of[specimen_id] = pool
        self._barcode_length = max(self._barcode_length, len(b1), len(b2))
        resolved = []
        for name, direction, barcode in ((p1, Primer.FWD, b1), (p2, Primer.REV, b2)):
            infos = self._resolve_primer_name(name, pool, direction)
            for info in infos:
                owner = self._primers.setdefault(info.primer, info)
                owner.add_barcode(barcode)
                owner.specimens.add(specimen_id)
            resolved.append(infos)
        self._specimens.append((specimen_id, pool, b1, resolved[0], b2, resolved[1]))
        self._primer_pairings.clear()

    def prune_unused_pools(self):
        unused = set(self._primer_registry.get_pools()) - self._active_pools
        if not unused:
            return
        logging.info(f"Removing unused pools: {unused}")
        for primer in self._primers.values():
            primer.pools = [p for p in primer.pools if p in self._active_pools]
        for pool in unused:
            self._primer_registry.drop_pool(pool)
        stats = self._primer_registry.get_pool_stats()
        logging.info(f"After pruning: {stats['total_primers']} primers in {stats['total_pools']} pools")

    def specimens_for_barcodes_and_primers(self, b1_list, b2_list, p1_matched, p2_matched) -> List[str]:
        return [sid for sid, _pool, b1, p1s, b2, p2s in self._specimens
                if p1_matched in p1s and p2_matched in p2s and b1.upper() in b1_list and b2.upper() in b2_list]

    def specimen_for_exact_match(self, b1: str, b2: str, p1: PrimerInfo, p2: PrimerInfo) -> Optional[str]:
        hits = self.specimens_for_barcodes_and_primers([b1.upper()], [b2.upper()], p1, p2)
        return hits[0] if hits else None

    def get_primers(self, direction: Primer) -> List[PrimerInfo]:
        return [p for p in self._primers.values() if p.direction == direction]

    def get_paired_primers(self, primer: str) -> List[PrimerInfo]:
        if primer not in self._primer_pairings:
            me = self._primers[primer]
            self._primer_pairings[primer] = [other for other in self._primers.values()
                                             if other.direction != me.direction and other.specimens & me.specimens]
        return self._primer_pairings[primer]

    def get_specimen_pool(self, specimen_id: str) -> Optional[str]:
        return self._pool_of.get(specimen_id)

    def b_length(self) -> int:
        return self._barcode_length

    def _barcode_sets(self):
        fwd, rev = [], []
        for primer in self._primers.values():
            dest = fwd if primer.direction == Primer.FWD else rev
            dest.extend(b for b in primer.barcodes if b not in dest)
        return fwd, rev

    def validate(self):
        fwd, rev = self._barcode_sets()
        dups = set(fwd) & set(rev)
        if dups:
            logging.warning(f"Duplicate Barcodes ({len(dups)}) in Fwd and Rev: {dups}")
        if len({len(b) for b in fwd}) > 1:
            logging.warning("Forward barcodes have inconsistent lengths")
        if len({len(b) for b in rev}) > 1:
            logging.warning("Reverse barcodes have inconsistent lengths")
        self.prune_unused_pools()


class BarcodePrefilter(Protocol):
    """Plugin protocol of the reference (databases.py:311-316).  The GPU path evaluates the Bloom
    prefilter's exact-set rule in-kernel; objects of this protocol only select on/off."""

    def match(self, barcode: str, sequence: str) -> bool: ...


class PassthroughPrefilter:
    """Prefilter that never rejects (== --disable-prefilter)."""

    def match(self, barcode: str, sequence: str) -> bool:
        return True
