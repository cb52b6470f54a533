"""MI355X counterpart of the reference's ``faiss_index_corpus.py`` (``build_faiss_index`` :27-52): read the pickled shards
``corpus_embeddings_*.pkl`` / ``passage_id_list_*.pkl`` in ``end``-index order, append them to an ``Indexer`` (resident in
HBM), serialise ``index.faiss`` + ``index_meta.faiss`` and delete the shard files — same flags, same side effects."""
from __future__ import annotations

import argparse
import glob
import logging
import os
import pickle

from .retriever.index import Indexer

logger = logging.getLogger(__file__)


def setup_parser(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--index_name", type=str, default="ip_indexer")
    parser.add_argument("--index_folder", type=str, default=None)
    parser.add_argument("--embedding_size", type=int, default=1024)
    return parser.parse_args(argv)


def _end_index(path: str) -> int:
    return int(os.path.basename(path).split(".")[0].split("_")[-1])


def sort_embedding_files(files):
    files.sort(key=_end_index)
    return files


def build_faiss_index(args, serialize: bool = True, remove_shards: bool = True) -> Indexer:
    indexer = Indexer(args.embedding_size, metric="inner_product")
    embedding_files = sort_embedding_files(glob.glob(os.path.join(args.index_folder, "corpus_embeddings_*.pkl")))
    id_files = {_end_index(p): p for p in glob.glob(os.path.join(args.index_folder, "passage_id_list_*.pkl"))}
    assert len(embedding_files) == len(id_files)
    total = 0
    pairs = []
    for f in embedding_files:
        pairs.append((f, id_files[_end_index(f)]))      # matched by the end index, as faiss_index_corpus.py:36-41 does
        total = max(total, _end_index(f) + 1)
    indexer.index.reserve(total)
    for emb_file, id_file in pairs:
        with open(emb_file, "rb") as fh:
            embeddings = pickle.load(fh)
        with open(id_file, "rb") as fh:
            passage_ids = pickle.load(fh)
        indexer.index_data(passage_ids, embeddings.cpu().numpy())
    if serialize:
        logger.info(f"Saving index to {args.index_folder} ... ")
        indexer.serialize(args.index_folder)
    if remove_shards:
        for f in embedding_files + list(id_files.values()):
            os.remove(f)
    return indexer


if __name__ == "__main__":
    build_faiss_index(setup_parser())
