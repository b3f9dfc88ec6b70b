"""Input side of the lookup path: Parquet -> values + row_splits in HBM
(host mirror of ``hybridbackend/tensorflow/data``)."""
from hybridbackend_amd.data.parquet import ParquetDataset
