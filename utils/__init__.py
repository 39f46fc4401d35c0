"""Import-path shim: `from utils.losses import ...` resolves to the MI355X-native losses."""
