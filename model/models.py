from transformertts_amd.model.models import ForwardTransformer  # noqa: F401
