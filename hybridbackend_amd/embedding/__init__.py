"""Embedding ops of the sharded lookup path
(host mirror of ``hybridbackend/tensorflow/embedding``)."""
from hybridbackend_amd.embedding import cache
from hybridbackend_amd.embedding.hierarchical import HierarchicalGroupLookup
from hybridbackend_amd.embedding.lookup import GroupLookup
from hybridbackend_amd.embedding.lookup import GroupLookupGrad
from hybridbackend_amd.embedding.lookup import group_lookup
from hybridbackend_amd.embedding.sharded import PipelinedLookup
from hybridbackend_amd.embedding.sharded import ShardedGroupLookup
from hybridbackend_amd.embedding.unique import UniqueN
from hybridbackend_amd.embedding.unique import unique
from hybridbackend_amd.embedding.unique import unique_n
from hybridbackend_amd.embedding.variables import allocate_tables
from hybridbackend_amd.embedding.variables import shard_of_table
from hybridbackend_amd.embedding.variables import sharded_bucket_size
