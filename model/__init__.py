"""Import-path shim: `from model.models import ForwardTransformer` / `from model.factory import
tts_custom` resolve to the MI355X-native implementation."""
