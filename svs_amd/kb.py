"""
Host-side mirror of the reference's ``KB`` / ``AsyncKB`` **for the retrieve()
path only** (SURVEY.md section 8(a) rows A1, A6, A7, A8).

The reference's Python files do not travel to the GPU box, so this module gives
the parity tests (and users without the reference installed) the same surface:
``KB(path, embedding_func)``, ``bulk_add_docs()``, ``bulk_del_docs()``,
``retrieve(query, n) -> List[Retrieval]``, ``close()``, ``len()`` -- same
names, argument meaning and error behaviour as reference src/svs/kb.py:1407-1640
(sync) and :925-1206 (async).  Storage is a plain SQLite file with the
reference's schema v1 table layout (src/svs/kb.py:64-113), so a database
written by either implementation opens in the other; the graph / key-value /
hierarchy features of the reference are NOT rebuilt (out of scope, pure SQL).

The similarity search itself (``superheavy()``, src/svs/kb.py:1622-1627) is
``DeviceEmbeddingsMatrix.search`` -> HIP.  There is no numpy fallback.
"""
from __future__ import annotations

import asyncio
import json
import logging
import sqlite3
import struct
import sys
import threading
from contextlib import asynccontextmanager, contextmanager
from typing import Any, Awaitable, Callable, Dict, Iterator, List, Optional, Tuple

import numpy as np

from .index import DeviceIndex
from .matrix import DeviceEmbeddingsMatrix

_LOG = logging.getLogger(__name__)

EmbeddingFunc = Callable[[List[str]], Awaitable[List[List[float]]]]   # reference src/svs/types.py:12

EMBEDDING_MAGNITUDE_TOLERANCE = 0.001   # reference src/svs/kb.py:58
BULK_EMBEDDING_CHUNK_SIZE = 200         # reference src/svs/kb.py:52
SCHEMA_VERSION = 1                      # reference src/svs/kb.py:61

# Table layout of the reference's schema v1 (data format, src/svs/kb.py:64-113).
_SCHEMA = [
    "CREATE TABLE IF NOT EXISTS keyval (id INTEGER PRIMARY KEY, key TEXT NOT NULL UNIQUE, val ANY NOT NULL) STRICT",
    "CREATE TABLE IF NOT EXISTS keyval_user (id INTEGER PRIMARY KEY, key TEXT NOT NULL UNIQUE, val ANY NOT NULL) STRICT",
    "CREATE TABLE IF NOT EXISTS embeddings (id INTEGER PRIMARY KEY, embedding BLOB NOT NULL) STRICT",
    "CREATE TABLE IF NOT EXISTS docs (id INTEGER PRIMARY KEY, parent_id INTEGER REFERENCES docs(id), "
    "level INTEGER NOT NULL, text TEXT NOT NULL, embedding INTEGER REFERENCES embeddings(id), meta TEXT) STRICT",
    "CREATE INDEX IF NOT EXISTS idx_docs_parent_id ON docs(parent_id)",
    "CREATE INDEX IF NOT EXISTS idx_docs_level ON docs(level)",
    "CREATE INDEX IF NOT EXISTS idx_docs_embedding ON docs(embedding)",
    "CREATE TABLE IF NOT EXISTS edges (id INTEGER PRIMARY KEY, a INTEGER REFERENCES docs(id) NOT NULL, "
    "b INTEGER REFERENCES docs(id) NOT NULL, r INTEGER REFERENCES docs(id) NOT NULL, w REAL, d INTEGER NOT NULL) STRICT",
    "CREATE UNIQUE INDEX IF NOT EXISTS idx_edges_abr ON edges(a, b, r)",
    "CREATE INDEX IF NOT EXISTS idx_edges_a ON edges(a)",
    "CREATE INDEX IF NOT EXISTS idx_edges_b ON edges(b)",
    "CREATE INDEX IF NOT EXISTS idx_edges_r ON edges(r)",
    "CREATE INDEX IF NOT EXISTS idx_edges_d ON edges(d)",
]


def embedding_to_bytes(embedding: List[float]) -> bytes:
    """Little-endian f32 BLOB codec, reference src/svs/embeddings/util.py:15-16."""
    return np.asarray(embedding, dtype="<f4").tobytes() if len(embedding) else b""


def embedding_from_bytes(blob: bytes) -> List[float]:
    """Reference src/svs/embeddings/util.py:19-23."""
    assert len(blob) % 4 == 0
    return [float(x) for x in np.frombuffer(blob, dtype="<f4")]


def check_magnitude(vectors: List[List[float]], tolerance: float = EMBEDDING_MAGNITUDE_TOLERANCE) -> None:
    """Unit-norm guard, reference src/svs/embeddings/util.py:34-39 (f32 norms).
    The HIP kernels compute a plain dot product and do NOT re-normalise, exactly
    like the reference, so this guard is what makes the score a cosine."""
    v = np.array(vectors, dtype=np.float32)
    if v.ndim != 2:
        raise ValueError("embedding function must return a list of equal-length vectors")
    mags = np.sqrt((v * v).sum(axis=1))
    if (np.abs(mags - 1.0) > tolerance).any():
        raise ValueError("embedding magnitude out of spec")


class _Store:
    """Minimal SQLite access for the retrieve() path.  One connection, used from
    one thread at a time (callers serialise, as the reference does with its
    asyncio.Lock -- src/svs/kb.py:942-945)."""

    def __init__(self, path: str):
        self.path = str(path)
        self.conn = sqlite3.connect(self.path, isolation_level=None, check_same_thread=False)
        self.conn.execute("PRAGMA foreign_keys = ON")
        self._tx = threading.RLock()
        with self.transaction():
            for stmt in _SCHEMA:
                self.conn.execute(stmt)
            row = self.conn.execute("SELECT val FROM keyval WHERE key = 'schema_version'").fetchone()
            if row is None:
                self.conn.execute("INSERT INTO keyval (key, val) VALUES ('schema_version', ?)", (SCHEMA_VERSION,))
            elif row[0] != SCHEMA_VERSION:
                raise RuntimeError(f"unsupported schema version {row[0]}")

    @contextmanager
    def transaction(self):
        """All-or-nothing, like the reference's ``with db as q`` (src/svs/kb.py:806-817)."""
        with self._tx:
            self.conn.execute("BEGIN")
            try:
                yield self
            except BaseException:
                self.conn.execute("ROLLBACK")
                raise
            else:
                self.conn.execute("COMMIT")

    def close(self) -> None:
        self.conn.close()

    # -- docs ---------------------------------------------------------------
    def add_doc(self, text: str, parent_id: Optional[int], meta: Optional[Dict[str, Any]]) -> int:
        level = 0
        if parent_id is not None:
            row = self.conn.execute("SELECT level FROM docs WHERE id = ?", (parent_id,)).fetchone()
            if row is None:
                raise ValueError(f"invalid parent_id: {parent_id}")
            level = row[0] + 1
        cur = self.conn.execute(
            "INSERT INTO docs (parent_id, level, text, embedding, meta) VALUES (?, ?, ?, NULL, ?)",
            (parent_id, level, text, json.dumps(meta) if meta is not None else None))
        return int(cur.lastrowid)

    def set_doc_embedding(self, doc_id: int, blob: bytes) -> int:
        cur = self.conn.execute("INSERT INTO embeddings (embedding) VALUES (?)", (blob,))
        res = self.conn.execute("UPDATE docs SET embedding = ? WHERE id = ?", (cur.lastrowid, doc_id))
        if res.rowcount != 1:
            raise KeyError(doc_id)
        return int(cur.lastrowid)

    def del_doc(self, doc_id: int) -> Optional[int]:
        """Returns the id of the embedding that went with the document (if any)."""
        row = self.conn.execute("SELECT embedding FROM docs WHERE id = ?", (doc_id,)).fetchone()
        if row is None:
            raise KeyError(doc_id)
        self.conn.execute("DELETE FROM docs WHERE id = ?", (doc_id,))
        if row[0] is not None:
            self.conn.execute("DELETE FROM embeddings WHERE id = ?", (row[0],))
        return row[0]

    def count_docs(self) -> int:
        return int(self.conn.execute("SELECT COUNT(*) FROM docs").fetchone()[0])

    def fetch_doc_for_embedding(self, emb_id: int) -> Dict[str, Any]:
        """The two SELECTs per result of reference src/svs/kb.py:1633-1634, as one
        join; the record has the shape of fetch_doc(include_embedding=False)
        (src/svs/kb.py:464-473)."""
        row = self.conn.execute(
            "SELECT id, parent_id, level, text, embedding, meta FROM docs WHERE embedding = ?", (emb_id,)).fetchone()
        if row is None:
            raise KeyError(emb_id)
        return {"id": row[0], "parent_id": row[1], "level": row[2], "text": row[3],
                "embedding": row[4] is not None, "meta": json.loads(row[5]) if row[5] is not None else None}

    def fetch_docs_for_embeddings(self, emb_ids: List[int]) -> Dict[int, Dict[str, Any]]:
        """One ``IN (...)`` query for a whole result list (with
        the search on the GPU, the reference's 2 SELECTs per result become the
        bottleneck of ``retrieve()``)."""
        out: Dict[int, Dict[str, Any]] = {}
        for c0 in range(0, len(emb_ids), 500):   # stay under SQLITE_MAX_VARIABLE_NUMBER
            chunk = emb_ids[c0:c0 + 500]
            marks = ",".join("?" * len(chunk))
            for row in self.conn.execute(
                    f"SELECT id, parent_id, level, text, embedding, meta FROM docs WHERE embedding IN ({marks})", chunk):
                out[row[4]] = {"id": row[0], "parent_id": row[1], "level": row[2], "text": row[3],
                               "embedding": True, "meta": json.loads(row[5]) if row[5] is not None else None}
        return out

    # -- A7: the producer of the matrix ---------------------------------------
    def build_embeddings_matrix(self) -> Tuple[np.ndarray, np.ndarray]:
        """Same contract as reference src/svs/kb.py:573-618 -- f32 (n, m) C-contiguous
        matrix + i64 ids, rows in ``SELECT id, embedding FROM embeddings`` order, m
        from the first row, empty table -> shape (0, 0) -- but each BLOB (already
        little-endian f32, src/svs/embeddings/util.py:15-16) is byte-copied straight
        into its row through a memoryview instead of ``struct.unpack`` -> python list
        -> element-wise assign: 4.7 s instead of ~130 s per 1M rows on the build
        container (the reference's published 98.7 s cold start)."""
        n = int(self.conn.execute("SELECT COUNT(*) FROM embeddings").fetchone()[0])
        first = self.conn.execute("SELECT embedding FROM embeddings LIMIT 1").fetchone()
        m = len(first[0]) // 4 if first is not None else 0
        matrix = np.zeros((n, m), dtype=np.float32)
        lookup = np.zeros(n, dtype=np.int64)
        assert sys.byteorder == "little"   # the BLOB codec is '<f'; a big-endian host would need a byteswap
        raw = memoryview(matrix).cast("B") if n * m else None
        rb = m * 4
        i = -1
        for i, (emb_id, blob) in enumerate(self.conn.execute("SELECT id, embedding FROM embeddings")):
            assert len(blob) == rb
            if rb:
                raw[i * rb:(i + 1) * rb] = blob
            lookup[i] = emb_id
        assert i == n - 1
        return matrix, lookup

    def embedding_blocks(self, block_bytes: int = 32 << 20, acquire=None):
        """The same rows as ``build_embeddings_matrix`` without the (n, m) host matrix: yields
        (n, m) first, then (ids i64[r], rows f32[r, m]) blocks of ~32 MiB -- the size of the library's
        pinned staging chunks -- each BLOB byte-copied into the block through a memoryview.  The
        consumer appends every block to the HBM copy (svs_index_append) before asking for the next,
        so the host never holds more than one block of a 6 GB (1M x 1536) corpus.  The block buffer is
        reused: copy what you keep.  With ``acquire`` every block is decoded into the (cap, m) f32
        array that call returns (asked for again after each yield) instead of a private buffer."""
        n = int(self.conn.execute("SELECT COUNT(*) FROM embeddings").fetchone()[0])
        first = self.conn.execute("SELECT embedding FROM embeddings LIMIT 1").fetchone()
        m = len(first[0]) // 4 if first is not None else 0
        yield n, m
        if n * m == 0:
            return
        assert sys.byteorder == "little"
        rb = m * 4

        def fresh():
            b = acquire() if acquire is not None else np.empty((max(1, block_bytes // rb), m), dtype=np.float32)
            assert b.dtype == np.float32 and b.ndim == 2 and b.shape[1] == m and b.flags["C_CONTIGUOUS"]
            return b, memoryview(b).cast("B")

        buf, raw = fresh()
        cap = buf.shape[0]
        ids = np.empty(cap, dtype=np.int64)
        fill = total = 0
        for emb_id, blob in self.conn.execute("SELECT id, embedding FROM embeddings"):
            assert len(blob) == rb
            raw[fill * rb:(fill + 1) * rb] = blob
            ids[fill] = emb_id
            fill += 1
            if fill == cap:
                yield ids[:fill], buf[:fill]
                total += fill
                fill = 0
                if acquire is not None:      # the consumer has committed the block: take the other one
                    buf, raw = fresh()
                    assert buf.shape[0] == cap
        if fill:
            yield ids[:fill], buf[:fill]
            total += fill
        assert total == n


def _build_from_store(store: _Store) -> Tuple[np.ndarray, np.ndarray]:
    with store.transaction():
        return store.build_embeddings_matrix()


def _blocks_from_store(store: _Store, acquire=None):
    """Streaming form of the cold start: first (n, m), then (ids, rows)
    blocks in ``SELECT id, embedding FROM embeddings`` order, inside one read transaction.
    ``acquire()`` (optional) supplies the block to decode into -- the library's pinned staging
    memory (DeviceIndex.staging_acquire) -- so a BLOB's bytes are copied exactly once on the host."""
    with store.transaction():
        yield from store.embedding_blocks(acquire=acquire)


class KB:
    """Sync knowledge base, retrieve() path only.  Mirrors reference
    src/svs/kb.py:1407-1640."""

    def __init__(self, local_path: str, embedding_func: Optional[EmbeddingFunc] = None,
                 device: int = 0, index_factory: Callable[..., Any] = DeviceIndex, dtype: str = "f32"):
        if embedding_func is None:
            raise RuntimeError("No embedding function. You must pass the embedding function you want to use.")
        self.embedding_func = embedding_func
        self.db: Optional[_Store] = _Store(local_path)
        if dtype != "f32":   # storage dtype of the HBM copy ("f16", "fp8"); f32 is the reference's arithmetic
            import functools
            index_factory = functools.partial(index_factory, dtype=dtype)
        self.embeddings_matrix = DeviceEmbeddingsMatrix(device=device, builder=_build_from_store,
                                                        index_factory=index_factory, keep_host_matrix=False,
                                                        block_builder=_blocks_from_store)
        self._loop = asyncio.new_event_loop()

    # -- embedding helpers (A8) -------------------------------------------------
    def _embed(self, texts: List[str]) -> List[List[float]]:
        awaitable = self.embedding_func(texts)
        assert asyncio.iscoroutine(awaitable)
        vectors = self._loop.run_until_complete(awaitable)
        check_magnitude(vectors)
        return vectors

    def __len__(self) -> int:
        assert self.db is not None
        with self.db.transaction():
            return self.db.count_docs()

    def load(self) -> None:
        """Reference ``load()``: build the matrix now instead of at the first
        retrieve() (src/svs/kb.py:964-967 async twin)."""
        assert self.db is not None
        self.embeddings_matrix.get_sync(self.db)

    def close(self) -> None:
        if self.db is not None:
            self.db.close()
            self.db = None
            self.embeddings_matrix.invalidate()
        if not self._loop.is_closed():
            self._loop.close()

    @contextmanager
    def bulk_add_docs(self) -> Iterator[Callable[..., int]]:
        """Reference src/svs/kb.py:1486-1524: docs are inserted inside ONE
        transaction, embeddings are fetched in chunks of 200 at exit, then the
        cached matrix is invalidated."""
        assert self.db is not None
        new_ids: List[int] = []
        new_vecs: List[List[float]] = []
        try:
            with self.db.transaction():
                live = True
                pending: List[Tuple[int, str]] = []

                def add_doc(text: str, parent_id: Optional[int] = None, meta: Optional[Dict[str, Any]] = None,
                            no_embedding: bool = False) -> int:
                    assert live, "You may not call this function outside of the context manager!"
                    doc_id = self.db.add_doc(text, parent_id, meta)
                    if not no_embedding:
                        pending.append((doc_id, text))
                    return doc_id

                try:
                    yield add_doc
                finally:
                    live = False
                for c0 in range(0, len(pending), BULK_EMBEDDING_CHUNK_SIZE):
                    chunk = pending[c0:c0 + BULK_EMBEDDING_CHUNK_SIZE]
                    vectors = self._embed([t for _, t in chunk])
                    for (doc_id, _), vec in zip(chunk, vectors):
                        new_ids.append(self.db.set_doc_embedding(doc_id, embedding_to_bytes(vec)))
                        new_vecs.append(vec)
        except BaseException:
            self.embeddings_matrix.invalidate()
            raise
        # committed: the reference drops the whole cached matrix here (kb.py:1523); the
        # HBM copy is instead extended in place (falls back to invalidate when not loaded)
        if new_ids:
            self.embeddings_matrix.append(np.array(new_vecs, dtype=np.float32), new_ids)

    @contextmanager
    def bulk_del_docs(self) -> Iterator[Callable[[int], None]]:
        """Reference src/svs/kb.py:1526-1542."""
        assert self.db is not None
        gone: List[int] = []
        try:
            with self.db.transaction():
                live = True

                def del_doc(doc_id: int) -> None:
                    assert live, "You may not call this function outside of the context manager!"
                    emb_id = self.db.del_doc(doc_id)
                    if emb_id is not None:
                        gone.append(emb_id)

                try:
                    yield del_doc
                finally:
                    live = False
        except BaseException:
            self.embeddings_matrix.invalidate()
            raise
        if gone:   # reference: invalidate() (kb.py:1541); here the rows are tombstoned in HBM
            self.embeddings_matrix.remove(gone)

    def retrieve(self, query: str, n: int) -> List[Dict[str, Any]]:
        """Reference src/svs/kb.py:1608-1640.  Same four log lines, same result
        shape (``Retrieval``: {'score': float, 'doc': DocumentRecord})."""
        _LOG.info(f"retrieving {n} documents with query string: {query}")
        assert self.db is not None
        self.embeddings_matrix.get_sync(self.db)
        query_vec = np.array(self._embed([query])[0], dtype=np.float32)
        _LOG.info("got embedding for query!")
        emb_ids = self.embeddings_matrix.search(query_vec, n)          # superheavy(): HIP
        _LOG.info(f"computed {self.embeddings_matrix.index.shape[0]} cosine similarities")
        with self.db.transaction():
            res = [{"score": score, "doc": self.db.fetch_doc_for_embedding(emb_id)} for score, emb_id in emb_ids]
        _LOG.info(f"retrieved top {n} documents")
        return res


    def document_top_pairwise_scores(self, n: int) -> List[Tuple[float, Dict[str, Any], Dict[str, Any]]]:
        """Reference src/svs/kb.py:1642-1671: the n most similar document pairs,
        [(score, doc_1, doc_2)].  M.M^T and the upper-triangle top-n run on the GPU
        (svs_index_top_pairs)."""
        assert self.db is not None
        self.embeddings_matrix.get_sync(self.db)
        n_docs = self.embeddings_matrix.index.shape[0]
        _LOG.info(f"computing pairwise similarity over {n_docs} documents")
        pairs = self.embeddings_matrix.top_pairs(n)
        _LOG.info(f"computed {n_docs * n_docs} pairwise cosine similarities")
        with self.db.transaction():
            docs = self.db.fetch_docs_for_embeddings(sorted({e for _, a, b in pairs for e in (a, b)}))
        _LOG.info(f"retrieved top {n} document pairs")
        return [(score, docs[a], docs[b]) for score, a, b in pairs]

    def retrieve_many(self, queries: List[str], n: int) -> List[List[Dict[str, Any]]]:
        """Batched ``retrieve()``: the queries are embedded
        in chunks of 200 like bulk_add_docs, searched together (the corpus is read
        once per 16 queries instead of once per query) and each result list is
        fetched with one SQL statement.  Element i equals ``retrieve(queries[i], n)``."""
        _LOG.info(f"retrieving {n} documents for each of {len(queries)} query strings")
        assert self.db is not None
        self.embeddings_matrix.get_sync(self.db)
        vecs: List[List[float]] = []
        for c0 in range(0, len(queries), BULK_EMBEDDING_CHUNK_SIZE):
            vecs.extend(self._embed(queries[c0:c0 + BULK_EMBEDDING_CHUNK_SIZE]))
        if not vecs:
            return []
        per_query = self.embeddings_matrix.search_many(np.array(vecs, dtype=np.float32), n)
        out: List[List[Dict[str, Any]]] = []
        with self.db.transaction():
            for emb_ids in per_query:
                docs = self.db.fetch_docs_for_embeddings([e for _, e in emb_ids])
                out.append([{"score": s, "doc": docs[e]} for s, e in emb_ids])
        return out


class AsyncKB:
    """Async twin (reference src/svs/kb.py:925-1206): the matrix is fetched under
    the lock, ``superheavy`` runs on an executor thread OUTSIDE the lock
    (:1184-1190), so concurrent retrieve() calls search the same handle
    concurrently -- the C ABI is re-entrant for exactly this."""

    def __init__(self, local_path: str, embedding_func: Optional[EmbeddingFunc] = None,
                 device: int = 0, index_factory: Callable[..., Any] = DeviceIndex, dtype: str = "f32"):
        if embedding_func is None:
            raise RuntimeError("No embedding function. You must pass the embedding function you want to use.")
        self.embedding_func = embedding_func
        self._path = local_path
        self.db: Optional[_Store] = None
        self._lock: Optional[asyncio.Lock] = None
        if dtype != "f32":   # storage dtype of the HBM copy ("f16", "fp8"); f32 is the reference's arithmetic
            import functools
            index_factory = functools.partial(index_factory, dtype=dtype)
        self.embeddings_matrix = DeviceEmbeddingsMatrix(device=device, builder=_build_from_store,
                                                        index_factory=index_factory, keep_host_matrix=False,
                                                        block_builder=_blocks_from_store)

    def _get_lock(self) -> asyncio.Lock:
        if self._lock is None:
            self._lock = asyncio.Lock()
        return self._lock

    async def _ensure_db(self) -> _Store:
        if self.db is None:
            loop = asyncio.get_running_loop()
            self.db = await loop.run_in_executor(None, lambda: _Store(self._path))
        return self.db

    async def _embed(self, texts: List[str]) -> List[List[float]]:
        vectors = await self.embedding_func(texts)
        check_magnitude(vectors)
        return vectors

    async def load(self) -> None:
        async with self._get_lock():
            db = await self._ensure_db()
            await self.embeddings_matrix.get(db)

    async def close(self) -> None:
        async with self._get_lock():
            if self.db is not None:
                self.db.close()
                self.db = None
            self.embeddings_matrix.invalidate()

    async def count(self) -> int:
        async with self._get_lock():
            db = await self._ensure_db()
            with db.transaction():
                return db.count_docs()

    @asynccontextmanager
    async def bulk_add_docs(self):
        async with self._get_lock():
            db = await self._ensure_db()
            pending: List[Tuple[int, str]] = []
            live = True
            db.conn.execute("BEGIN")
            try:
                async def add_doc(text: str, parent_id: Optional[int] = None,
                                  meta: Optional[Dict[str, Any]] = None, no_embedding: bool = False) -> int:
                    assert live, "You may not call this function outside of the context manager!"
                    doc_id = db.add_doc(text, parent_id, meta)
                    if not no_embedding:
                        pending.append((doc_id, text))
                    return doc_id
                try:
                    yield add_doc
                finally:
                    live = False
                new_ids: List[int] = []
                new_vecs: List[List[float]] = []
                for c0 in range(0, len(pending), BULK_EMBEDDING_CHUNK_SIZE):
                    chunk = pending[c0:c0 + BULK_EMBEDDING_CHUNK_SIZE]
                    vectors = await self._embed([t for _, t in chunk])
                    for (doc_id, _), vec in zip(chunk, vectors):
                        new_ids.append(db.set_doc_embedding(doc_id, embedding_to_bytes(vec)))
                        new_vecs.append(vec)
            except BaseException:
                db.conn.execute("ROLLBACK")
                self.embeddings_matrix.invalidate()
                raise
            else:
                db.conn.execute("COMMIT")
            if new_ids:   # extend the HBM copy in place (reference: invalidate(), kb.py:1062)
                loop = asyncio.get_running_loop()
                await loop.run_in_executor(None, lambda: self.embeddings_matrix.append(
                    np.array(new_vecs, dtype=np.float32), new_ids))

    @asynccontextmanager
    async def bulk_del_docs(self):
        async with self._get_lock():
            db = await self._ensure_db()
            live = True
            gone: List[int] = []
            db.conn.execute("BEGIN")
            try:
                async def del_doc(doc_id: int) -> None:
                    assert live, "You may not call this function outside of the context manager!"
                    emb_id = db.del_doc(doc_id)
                    if emb_id is not None:
                        gone.append(emb_id)
                try:
                    yield del_doc
                finally:
                    live = False
            except BaseException:
                db.conn.execute("ROLLBACK")
                self.embeddings_matrix.invalidate()
                raise
            else:
                db.conn.execute("COMMIT")
            if gone:      # tombstone in HBM (reference: invalidate(), kb.py:1086)
                self.embeddings_matrix.remove(gone)

    async def retrieve(self, query: str, n: int) -> List[Dict[str, Any]]:
        """Reference src/svs/kb.py:1171-1206."""
        _LOG.info(f"retrieving {n} documents with query string: {query}")
        loop = asyncio.get_running_loop()
        async with self._get_lock():
            db = await self._ensure_db()
            await self.embeddings_matrix.get(db)
            # own a reference now: a later invalidate() must not affect this search
            idx, lookup, co = self.embeddings_matrix.hold_search()
        try:
            query_vec = np.array((await self._embed([query]))[0], dtype=np.float32)
            _LOG.info("got embedding for query!")

            def superheavy() -> List[Tuple[float, int]]:
                # tasks whose searches are in flight together share corpus passes (svs_amd/coalesce.py)
                res = co.search(idx, query_vec, n) if co is not None else idx.search(query_vec, n)
                return [(score, int(lookup.arr[row])) for score, row in res]

            emb_ids = await loop.run_in_executor(None, superheavy)
            _LOG.info(f"computed {idx.shape[0]} cosine similarities")
        finally:
            idx.release()
        async with self._get_lock():
            db = await self._ensure_db()

            def heavy() -> List[Dict[str, Any]]:
                with db.transaction():
                    return [{"score": s, "doc": db.fetch_doc_for_embedding(e)} for s, e in emb_ids]

            res = await loop.run_in_executor(None, heavy)
        _LOG.info(f"retrieved top {n} documents")
        return res

    async def retrieve_many(self, queries: List[str], n: int) -> List[List[Dict[str, Any]]]:
        """Async twin of ``KB.retrieve_many``.  Same structure as
        ``retrieve`` (reference src/svs/kb.py:1171-1206): the matrix is fetched under the lock, the
        batched search runs on an executor thread OUTSIDE it on a held reference, and every result
        list is fetched with one ``IN (...)`` statement.  Element i equals ``retrieve(queries[i], n)``."""
        _LOG.info(f"retrieving {n} documents for each of {len(queries)} query strings")
        loop = asyncio.get_running_loop()
        async with self._get_lock():
            db = await self._ensure_db()
            await self.embeddings_matrix.get(db)
            idx, lookup = self.embeddings_matrix.hold()
        try:
            vecs: List[List[float]] = []
            for c0 in range(0, len(queries), BULK_EMBEDDING_CHUNK_SIZE):
                vecs.extend(await self._embed(queries[c0:c0 + BULK_EMBEDDING_CHUNK_SIZE]))
            if not vecs:
                return []
            _LOG.info("got embeddings for the queries!")
            qmat = np.array(vecs, dtype=np.float32)

            def superheavy() -> List[List[Tuple[float, int]]]:
                scores, rows = idx.search_batch(qmat, n)
                arr = lookup.arr
                return [[(float(s), int(arr[r])) for s, r in zip(scores[i], rows[i])] for i in range(len(scores))]

            per_query = await loop.run_in_executor(None, superheavy)
            _LOG.info(f"computed {idx.shape[0]} x {len(vecs)} cosine similarities")
        finally:
            idx.release()
        async with self._get_lock():
            db = await self._ensure_db()

            def heavy() -> List[List[Dict[str, Any]]]:
                out: List[List[Dict[str, Any]]] = []
                with db.transaction():
                    for emb_ids in per_query:
                        docs = db.fetch_docs_for_embeddings([e for _, e in emb_ids])
                        out.append([{"score": s, "doc": docs[e]} for s, e in emb_ids])
                return out

            res = await loop.run_in_executor(None, heavy)
        _LOG.info(f"retrieved top {n} documents for {len(queries)} queries")
        return res

    async def document_top_pairwise_scores(self, n: int) -> List[Tuple[float, Dict[str, Any], Dict[str, Any]]]:
        """Reference src/svs/kb.py:1208-1243 (async twin of :1642-1671): the n most similar document
        pairs, [(score, doc_1, doc_2)].  ``superheavy`` -- there ``np.dot(M, M.T)`` + ``get_top_pairs``
        (src/svs/kb.py:1219, src/svs/util.py:206-233) -- is svs_index_top_pairs on an executor
        thread, outside the lock, on a held reference."""
        loop = asyncio.get_running_loop()
        async with self._get_lock():
            db = await self._ensure_db()
            await self.embeddings_matrix.get(db)
            idx, lookup = self.embeddings_matrix.hold()
        try:
            n_docs = idx.shape[0]
            _LOG.info(f"computing pairwise similarity over {n_docs} documents")

            def superheavy() -> List[Tuple[float, int, int]]:
                res = idx.top_pairs(n)
                arr = lookup.arr
                return [(score, int(arr[i]), int(arr[j])) for score, i, j in res]

            pairs = await loop.run_in_executor(None, superheavy)
            _LOG.info(f"computed {n_docs * n_docs} pairwise cosine similarities")
        finally:
            idx.release()
        async with self._get_lock():
            db = await self._ensure_db()

            def heavy() -> Dict[int, Dict[str, Any]]:
                with db.transaction():
                    return db.fetch_docs_for_embeddings(sorted({e for _, a, b in pairs for e in (a, b)}))

            docs = await loop.run_in_executor(None, heavy)
        _LOG.info(f"retrieved top {n} document pairs")
        return [(score, docs[a], docs[b]) for score, a, b in pairs]
