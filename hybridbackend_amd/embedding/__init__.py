"""Embedding ops of the sharded lookup path
(host mirror of ``hybridbackend/tensorflow/embedding``)."""
from hybridbackend_amd.embedding.lookup import GroupLookup
from hybridbackend_amd.embedding.lookup import group_lookup
