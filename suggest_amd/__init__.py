"""suggest_amd — MI355X-native engine for suggest-go's top-k fuzzy string matching path.

Host mirror of the reference's pkg/suggest API over libsuggest_hip.so (hand-written HIP for gfx950).
"""
from .index import IndexDescription, NGramIndex, pack_strings  # noqa: F401
from .metric import CosineMetric, DiceMetric, ExactMetric, JaccardMetric, OverlapMetric  # noqa: F401
from .service import ResultItem, SearchConfig, Service, read_configs, read_dictionary  # noqa: F401
from .spell import LanguageModel, SpellChecker  # noqa: F401
